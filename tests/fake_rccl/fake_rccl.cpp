// fake_rccl.cpp -- TEST INFRASTRUCTURE, never shipped: a host-side stand-in for the RCCL entry points that
// summertts_amd/csrc/multi.hip resolves at run time (ncclCommInitAll, ncclCommDestroy, ncclCommAbort, ncclAllGather, ncclSend,
// ncclRecv, ncclGroupStart, ncclGroupEnd, ncclGetErrorString), so that the N > 1 gather protocol of sts_multi -- pairing of sends
// and receives, zero-count ranks, a failing shard, buffer regrowth, the "rank 0 cannot receive" cancellation -- can be driven
// with THREE ranks on a box that has ONE GPU.  Every "rank" is a host thread of the same process on the same
// device; a collective is a host rendezvous + device-to-device copies.  Semantics kept: calls are stream-ordered (the caller's
// stream is drained before its buffers are touched and the copies are complete when the call returns -- stricter than RCCL, which
// is asynchronous), ncclSend blocks until the matching ncclRecv has taken the data, operations inside a group are deferred
// to ncclGroupEnd, and a communicator that was aborted fails every pending and later call instead of hanging.
// Selected through sts_multi_set_rccl_library(path, 1) by tests/test_parity_gpu.py only.  Fault injection: the environment
// variable FAKE_RCCL_FAIL_RECV=<n> makes the n-th ncclRecv of the process return ncclInternalError.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <stdlib.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <vector>

namespace {
struct World {
    std::mutex mu;
    std::condition_variable cv;
    int n = 0;
    bool aborted = false;
    // reusable barrier of the all-gather (every rank calls the collectives in the same order)
    int bar_count = 0; long bar_gen = 0;
    std::vector<const void*> ag_src;
    // mailboxes [src * n + dst]: a posted send waits here for its receive
    struct Msg { const void* ptr = nullptr; size_t bytes = 0; bool posted = false, taken = false; };
    std::vector<Msg> box;
    int refs = 0;
};
struct Comm { World* w; int rank; };
struct Deferred { bool send; void* ptr; size_t count; ncclDataType_t dt; int peer; Comm* c; hipStream_t st; };
thread_local int g_group_depth = 0;
thread_local std::vector<Deferred> g_deferred;
std::atomic<long> g_recv_calls{0};

size_t dt_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        default: return 8;
    }
}
template <typename P> bool wait_for(World* w, std::unique_lock<std::mutex>& lk, P pred) {
    w->cv.wait(lk, [&] { return w->aborted || pred(); });
    return !w->aborted;
}

ncclResult_t do_send(Comm* c, const void* buf, size_t count, ncclDataType_t dt, int peer, hipStream_t st) {
    World* w = c->w;
    if (peer < 0 || peer >= w->n || peer == c->rank) return ncclInvalidArgument;
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;          // the data is ready
    std::unique_lock<std::mutex> lk(w->mu);
    World::Msg& m = w->box[(size_t)c->rank * w->n + peer];
    if (!wait_for(w, lk, [&] { return !m.posted; })) return ncclInternalError;          // one message per (src, dst) in flight
    m.ptr = buf; m.bytes = count * dt_bytes(dt); m.posted = true; m.taken = false;
    w->cv.notify_all();
    if (!wait_for(w, lk, [&] { return m.taken; })) return ncclInternalError;            // the receiver has copied it
    m.posted = false;
    w->cv.notify_all();
    return ncclSuccess;
}
ncclResult_t do_recv(Comm* c, void* buf, size_t count, ncclDataType_t dt, int peer, hipStream_t st) {
    World* w = c->w;
    if (peer < 0 || peer >= w->n || peer == c->rank) return ncclInvalidArgument;
    std::unique_lock<std::mutex> lk(w->mu);
    World::Msg& m = w->box[(size_t)peer * w->n + c->rank];
    if (!wait_for(w, lk, [&] { return m.posted && !m.taken; })) return ncclInternalError;
    const size_t bytes = count * dt_bytes(dt);
    if (bytes != m.bytes) { w->aborted = true; w->cv.notify_all(); return ncclInvalidArgument; }      // a pairing bug: fail everybody loudly
    const void* src = m.ptr;
    lk.unlock();
    if (hipMemcpyAsync(buf, src, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    lk.lock();
    m.taken = true;
    w->cv.notify_all();
    return ncclSuccess;
}
}  // namespace

extern "C" {

ncclResult_t ncclCommInitAll(ncclComm_t* comms, int ndev, const int* devlist) {
    (void)devlist;                                   // (every emulated rank lives on whatever device its caller has made current)
    if (!comms || ndev < 1) return ncclInvalidArgument;
    World* w = new World();
    w->n = ndev; w->ag_src.assign(ndev, nullptr); w->box.assign((size_t)ndev * ndev, World::Msg()); w->refs = ndev;
    for (int r = 0; r < ndev; r++) comms[r] = (ncclComm_t) new Comm{w, r};
    return ncclSuccess;
}
static void release(Comm* c) {
    World* w = c->w;
    bool last;
    { std::lock_guard<std::mutex> lk(w->mu); last = --w->refs == 0; }
    delete c;
    if (last) delete w;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) { if (comm) release((Comm*)comm); return ncclSuccess; }
ncclResult_t ncclCommAbort(ncclComm_t comm) {
    if (!comm) return ncclSuccess;
    Comm* c = (Comm*)comm;
    { std::lock_guard<std::mutex> lk(c->w->mu); c->w->aborted = true; }
    c->w->cv.notify_all();
    // (nothing is freed: peer threads may still be inside a call on this world -- a test library can afford the leak)
    return ncclSuccess;
}
ncclResult_t ncclCommCount(const ncclComm_t comm, int* count) {
    if (!comm || !count) return ncclInvalidArgument;
    *count = ((Comm*)comm)->w->n;
    return ncclSuccess;
}
const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : (r == ncclInvalidArgument ? "fake rccl: invalid argument / unmatched transfer" : "fake rccl: aborted or injected failure"); }

static bool barrier(World* w, std::unique_lock<std::mutex>& lk) {
    const long gen = w->bar_gen;
    if (++w->bar_count == w->n) { w->bar_count = 0; w->bar_gen++; w->cv.notify_all(); return !w->aborted; }
    return wait_for(w, lk, [&] { return w->bar_gen != gen; });
}
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t dt, ncclComm_t comm, hipStream_t st) {
    Comm* c = (Comm*)comm;
    World* w = c->w;
    const size_t bytes = sendcount * dt_bytes(dt);
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;          // this rank's contribution is ready
    std::unique_lock<std::mutex> lk(w->mu);
    w->ag_src[c->rank] = sendbuff;
    if (!barrier(w, lk)) return ncclInternalError;                                      // every rank has posted its buffer
    std::vector<const void*> src = w->ag_src;
    lk.unlock();
    bool ok = true;
    for (int r = 0; r < w->n && ok; r++)
        ok = hipMemcpyAsync((char*)recvbuff + (size_t)r * bytes, src[r], bytes, hipMemcpyDeviceToDevice, st) == hipSuccess;
    ok = ok && hipStreamSynchronize(st) == hipSuccess;
    lk.lock();
    if (!barrier(w, lk)) return ncclInternalError;                                      // nobody reuses its send buffer before all have copied
    return ok ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t ncclGroupStart() { g_group_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
    if (g_group_depth <= 0) return ncclInvalidUsage;
    if (--g_group_depth > 0) return ncclSuccess;
    std::vector<Deferred> ops;
    ops.swap(g_deferred);
    ncclResult_t rc = ncclSuccess;
    for (const Deferred& d : ops) {
        const ncclResult_t r = d.send ? do_send(d.c, d.ptr, d.count, d.dt, d.peer, d.st) : do_recv(d.c, d.ptr, d.count, d.dt, d.peer, d.st);
        if (r != ncclSuccess && rc == ncclSuccess) rc = r;
    }
    return rc;
}
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
    if (g_group_depth > 0) { g_deferred.push_back(Deferred{true, (void*)sendbuff, count, dt, peer, (Comm*)comm, st}); return ncclSuccess; }
    return do_send((Comm*)comm, sendbuff, count, dt, peer, st);
}
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t dt, int peer, ncclComm_t comm, hipStream_t st) {
    const char* f = getenv("FAKE_RCCL_FAIL_RECV");
    if (f && g_recv_calls.fetch_add(1) + 1 == atol(f)) return ncclInternalError;
    if (g_group_depth > 0) { g_deferred.push_back(Deferred{false, recvbuff, count, dt, peer, (Comm*)comm, st}); return ncclSuccess; }
    return do_recv((Comm*)comm, recvbuff, count, dt, peer, st);
}

}  // extern "C"
