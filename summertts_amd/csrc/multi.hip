// multi.hip -- one host process, several MI355X: the multi-device form of sts_infer_ids_batch (SURVEY.md 8b "+ multi-device
// variant taking a device list", 8e).  No reference counterpart: SynthesizerTrn::infer handles one utterance on the CPU
// (/root/reference/src/models/SynthesizerTrn.cpp:323).
//
// Utterances are independent, so the batch is sharded BY UTTERANCE with no exchange step: weights are replicated (one
// engine = one stream set + ~116 MB of repacked weights per device), the shards are balanced longest-first by phoneme
// count (the same rule as summertts_amd/sharding.py: work is ~proportional to it), every device runs its shard as ONE
// packed variable-length batch on its own worker thread, downloads its int16 PCM through its own pinned staging buffer,
// and the caller gets the utterances back in INPUT order.  The result lives on the host (malloc'd per utterance, like
// sts_infer_ids_batch).
//
// Two ways of bringing the PCM home (sts_multi_gather_mode):
//   * per-device download (the default, STS_MULTI_AUTO / STS_MULTI_DOWNLOAD): each worker downloads its own shard over PCIe.
//   * RCCL gather (STS_MULTI_RCCL, opt-in; the xGMI path BASELINE.json's north_star names): the handle owns one RCCL communicator
//     per device (ncclCommInitAll; distinct devices).  After its shard has run, every worker publishes its sample count with
//     ncclAllGather; rank 0 sizes the gather buffer and every rank learns through a second one-word ncclAllGather whether
//     rank 0 can receive (a local failure on ANY rank before the transfer therefore cancels the transfer on ALL ranks instead of
//     leaving peers blocked in ncclSend); then ranks > 0 ncclSend their int16 PCM straight out of the engine's device buffer, rank 0
//     posts the matching ncclRecvs (one group) into the gather buffer on device 0 laid out by sts_multi_gather_layout, adds its own
//     shard device-to-device, and ONE device-to-host copy brings the whole batch down.  Every wait on a collective is bounded
//     (kCollectiveTimeoutMs): on a timeout or an RCCL error all communicators are aborted (ncclCommAbort), the call fails and
//     the handle continues with per-device downloads.  librccl is dlopen'ed on first use: a process that never asks for the gather
//     never loads it.
// The RCCL path has been exercised with one rank on the real library and with THREE ranks against tests/fake_rccl (a host-side
// stand-in for the seven entry points used here, selected through sts_multi_set_rccl_library -- a test hook): pairing, zero-count
// ranks, a failing shard, buffer regrowth.  Hardware N > 1 remains unmeasured (no multi-GPU node was available to the builder) --
// which is why the automatic mode does not select it.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>

#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "engine.hpp"

using namespace sts;

namespace {
// ---- librccl, resolved at run time
struct Rccl {
    void* h = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;          // optional
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;   // optional (sts_multi_rccl_ranks)
    std::string err;
    std::string override_path; bool allow_repeated = false;   // sts_multi_set_rccl_library (test hook)
    bool load() {
        if (h) return true;
        if (!override_path.empty()) h = dlopen(override_path.c_str(), RTLD_NOW | RTLD_LOCAL);
        else for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { h = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (h) break; }
        if (!h) { err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return false; }
        auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) err = std::string("librccl lacks ") + n; return p; };
        CommInitAll = (decltype(CommInitAll))sym("ncclCommInitAll"); CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        AllGather = (decltype(AllGather))sym("ncclAllGather"); Send = (decltype(Send))sym("ncclSend"); Recv = (decltype(Recv))sym("ncclRecv");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart"); GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        CommAbort = (decltype(CommAbort))dlsym(h, "ncclCommAbort");
        CommCount = (decltype(CommCount))dlsym(h, "ncclCommCount");
        return CommInitAll && CommDestroy && AllGather && Send && Recv && GroupStart && GroupEnd && GetErrorString;
    }
};
Rccl& rccl() { static Rccl r; return r; }
std::mutex g_rccl_mu;

struct Shard {
    std::vector<int> utt;              // indices into the caller's batch, ascending
    std::vector<int16_t> pcm;          // packed PCM of the shard (utterance order of `utt`)
    std::vector<int32_t> n_samples;
    int rc = STS_OK; std::string err;
};
}  // namespace

struct sts_multi {
    std::vector<std::unique_ptr<Engine>> engines;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    // one job at a time (the entry points are not re-entrant, like an engine)
    int64_t epoch = 0; int pending = 0; bool stop = false;
    int32_t B = 0; const int32_t* const* ids = nullptr; const int32_t* n = nullptr; const int32_t* sid = nullptr; const float* ls = nullptr;
    std::vector<Shard> shards;
    // RCCL gather state (gather_mode == 1)
    int gather_mode = 0;
    std::vector<ncclComm_t> comms;
    std::vector<long long*> d_counts;       // per device: [1 own word | ndev gathered words]
    int16_t* d_gather = nullptr; size_t gather_cap = 0;      // on engines[0]'s device
    int16_t* h_gather = nullptr; size_t h_gather_cap = 0;    // pinned
    std::vector<int64_t> counts, offsets; int64_t gather_total = 0;
    bool rccl_broken = false;               // a collective failed or timed out: communicators aborted, downloads from the next call on
    double last_gather_ms = 0;              // rank 0's wall time inside the last call's gather: counts -> ready round -> transfers -> the one download
    static constexpr int kCollectiveTimeoutMs = 60000;

    // Who may touch comms[k] (round 5): every RCCL host call of rank k goes through rccl_call(k, ...), which takes the
    // communicator under `mu`, marks the rank as inside a call, and refuses to start once the handle is broken.  abort_all() -- any rank's
    // failure -- brings down, under the same lock, every communicator whose owner is NOT inside a host call (ranks parked in wait_stream:
    // the aborted collective's kernel leaves their stream) and marks the others; an owner that was inside a call aborts its own
    // communicator when the call returns.  So no thread ever passes an aborted (freed) or null communicator into RCCL.
    std::vector<int> in_call; std::vector<char> abort_pending;
    int gather_arrived = 0; int64_t gather_epoch = 0;       // host barrier in front of the gather (the collective timeout must not cover the peers' inference)
    static constexpr int kAbortGraceMs = 2000;
    void abort_all() {
        bool pending = false;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (rccl_broken) return;
            rccl_broken = true;
            Rccl& R = rccl();
            for (size_t k = 0; k < comms.size(); k++) {
                if (!comms[k]) continue;
                if (in_call[k] > 0) { abort_pending[k] = 1; pending = true; continue; }
                if (R.CommAbort) (void)R.CommAbort(comms[k]);
                comms[k] = nullptr;       // (without ncclCommAbort in the library the communicator is simply abandoned)
            }
        }
        if (!pending) return;
        // Watchdog: an owner inside an RCCL host call normally returns at once (the peers' communicators are gone) and aborts its
        // own.  A host call that BLOCKS on the failed peer (lazy connection set-up inside the first ncclAllGather, ncclGroupEnd) never returns by
        // itself: after a grace period the aborting thread calls ncclCommAbort on it from here -- RCCL's documented way to unblock a stuck call.
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
            {
                std::lock_guard<std::mutex> lk(mu);
                bool any = false;
                for (size_t k = 0; k < comms.size(); k++) any = any || (abort_pending[k] && comms[k]);
                if (!any) return;
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(kAbortGraceMs)) {
                    Rccl& R = rccl();
                    for (size_t k = 0; k < comms.size(); k++)
                        if (abort_pending[k] && comms[k]) { if (R.CommAbort) (void)R.CommAbort(comms[k]); comms[k] = nullptr; abort_pending[k] = 0; }
                    return;
                }
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    bool broken() { std::lock_guard<std::mutex> lk(mu); return rccl_broken; }
    template <typename F> ncclResult_t rccl_call(int k, F&& f) {
        ncclComm_t c;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (rccl_broken || !comms[k]) return ncclInternalError;
            in_call[k]++; c = comms[k];
        }
        const ncclResult_t r = f(c);
        {
            std::lock_guard<std::mutex> lk(mu);
            if (--in_call[k] == 0 && abort_pending[k]) {
                abort_pending[k] = 0;
                if (comms[k]) { if (rccl().CommAbort) (void)rccl().CommAbort(comms[k]); comms[k] = nullptr; }
            }
        }
        return r;
    }
    // bounded wait for everything queued on a rank's stream (collectives included); false also when a PEER brought the communicators down
    bool wait_stream(Shard& sh, hipStream_t st, const char* what) {
        const bool done = wait_stream_raw(sh, st, what);
        if (done && broken()) { if (sh.rc == STS_OK) { sh.rc = STS_EDEVICE; sh.err = std::string(what) + ": the communicators were aborted by a peer's failure"; } return false; }
        return done;
    }
    bool wait_stream_raw(Shard& sh, hipStream_t st, const char* what) {
        const auto t0 = std::chrono::steady_clock::now();
        for (long spin = 0;; spin++) {
            const hipError_t q = hipStreamQuery(st);
            if (q == hipSuccess) return true;
            if (q != hipErrorNotReady) { if (sh.rc == STS_OK) { sh.rc = STS_EDEVICE; sh.err = std::string(what) + ": " + hipGetErrorString(q); } abort_all(); return false; }
            if (spin > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50)); else std::this_thread::yield();
            if ((spin & 0xff) == 0xff && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(kCollectiveTimeoutMs)) {
                if (sh.rc == STS_OK) { sh.rc = STS_EDEVICE; sh.err = std::string(what) + ": timed out waiting for the peers (communicators aborted)"; }
                abort_all();
                (void)hipStreamSynchronize(st);     // the aborted collective's kernel leaves the stream
                return false;
            }
        }
    }

    // after the shard has run: counts -> all ranks, "can receive" of rank 0 -> all ranks, PCM of ranks > 0 -> rank 0 over xGMI,
    // rank 0 downloads everything at once.  No rank ever blocks on a transfer that another rank will not post.
    void rccl_gather(int k) {
        Shard& sh = shards[k];
        Engine& eng = *engines[k];
        Rccl& R = rccl();
        const int nd = (int)engines.size();
        const auto tg0 = std::chrono::steady_clock::now();
        struct GatherClock { sts_multi* m; int k; std::chrono::steady_clock::time_point t0;
                             ~GatherClock() { if (k == 0) m->last_gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } gclock{this, k, tg0};
        const long long mine = sh.rc == STS_OK && !sh.utt.empty() ? (long long)eng.total_samples : 0;    // a failed shard still takes part
        auto ck = [&](ncclResult_t r, const char* what) { if (r != ncclSuccess) { if (sh.rc == STS_OK) { sh.rc = STS_EDEVICE; sh.err = std::string(what) + ": " + (broken() ? "the communicators were aborted by a peer's failure" : R.GetErrorString(r)); } abort_all(); } return r == ncclSuccess; };
        auto hk = [&](hipError_t e, const char* what) { if (e != hipSuccess && sh.rc == STS_OK) { sh.rc = STS_EDEVICE; sh.err = std::string(what) + ": " + hipGetErrorString(e); } return e == hipSuccess; };
        long long host_words[65];
        // ---- round 1: sample counts
        bool ok = hk(hipMemcpyAsync(d_counts[k], &mine, sizeof(long long), hipMemcpyHostToDevice, eng.stream), "count upload");
        ok = ok && ck(rccl_call(k, [&](ncclComm_t cm) { return R.AllGather(d_counts[k], d_counts[k] + 1, sizeof(long long), ncclInt8, cm, eng.stream); }), "ncclAllGather (counts)");
        ok = ok && hk(hipMemcpyAsync(host_words, d_counts[k] + 1, sizeof(long long) * nd, hipMemcpyDeviceToHost, eng.stream), "count download");
        ok = ok && wait_stream(sh, eng.stream, "count exchange");
        if (!ok) { abort_all(); return; }                    // (a local HIP failure: nobody waits for this rank once the communicators are down)
        // ---- rank 0 sizes its buffers; round 2: every rank's "ready" word (rank 0: the buffers exist; others: nothing failed locally)
        long long ready = 1;
        if (k == 0) {
            counts.assign(host_words, host_words + nd);
            offsets.resize(nd);
            gather_total = sts_multi_gather_layout(counts.data(), nd, offsets.data());
            const size_t need = (size_t)std::max<int64_t>(gather_total, 1);
            if (need > gather_cap) {
                if (d_gather) (void)hipFree(d_gather);
                d_gather = nullptr; gather_cap = 0;
                if (hk(hipMalloc((void**)&d_gather, (need + need / 4) * 2), "gather buffer")) gather_cap = need + need / 4;
            }
            if (need > h_gather_cap) {
                if (h_gather) (void)hipHostFree(h_gather);
                h_gather = nullptr; h_gather_cap = 0;
                if (hk(hipHostMalloc((void**)&h_gather, (need + need / 4) * 2, hipHostMallocDefault), "pinned gather buffer")) h_gather_cap = need + need / 4;
            }
            ready = d_gather && h_gather && gather_cap >= need && h_gather_cap >= need ? 1 : 0;
        }
        ok = hk(hipMemcpyAsync(d_counts[k], &ready, sizeof(long long), hipMemcpyHostToDevice, eng.stream), "ready upload");
        ok = ok && ck(rccl_call(k, [&](ncclComm_t cm) { return R.AllGather(d_counts[k], d_counts[k] + 1, sizeof(long long), ncclInt8, cm, eng.stream); }), "ncclAllGather (ready)");
        ok = ok && hk(hipMemcpyAsync(host_words, d_counts[k] + 1, sizeof(long long) * nd, hipMemcpyDeviceToHost, eng.stream), "ready download");
        ok = ok && wait_stream(sh, eng.stream, "ready exchange");
        if (!ok) { abort_all(); return; }
        for (int p = 0; p < nd; p++)
            if (host_words[p] != 1) {                        // somebody cannot take part: no rank posts a transfer, the call fails, nothing hangs
                if (sh.rc == STS_OK) { sh.rc = STS_EDEVICE; sh.err = "RCCL gather cancelled: rank " + std::to_string(p) + " could not allocate its buffers"; }
                return;
            }
        // ---- the transfer (out of HBM: an engine of an RCCL handle never runs with host_pcm, so a host-resident PCM cannot occur here)
        if (mine > 0 && !eng.pcm_hbm()) { if (sh.rc == STS_OK) { sh.rc = STS_ESTATE; sh.err = "RCCL gather: the shard's PCM is not in device memory"; } abort_all(); return; }
        if (k > 0) {
            if (mine > 0 && ck(rccl_call(k, [&](ncclComm_t cm) { return R.Send(eng.pcm_hbm(), (size_t)mine, ncclHalf, 0, cm, eng.stream); }), "ncclSend"))
                (void)wait_stream(sh, eng.stream, "ncclSend");      // the engine's PCM buffer is free again
            return;
        }
        // (the receives of all peers form ONE group = one host call on rank 0's communicator)
        const char* what = "ncclGroupStart";
        const bool posted = ck(rccl_call(0, [&](ncclComm_t cm) {
            ncclResult_t r = R.GroupStart();
            if (r != ncclSuccess) return r;
            ncclResult_t first = ncclSuccess;
            for (int p = 1; p < nd && first == ncclSuccess; p++)
                if (counts[p] > 0) { first = R.Recv(d_gather + offsets[p], (size_t)counts[p], ncclHalf, p, cm, eng.stream); if (first != ncclSuccess) what = "ncclRecv"; }
            r = R.GroupEnd();
            if (first == ncclSuccess && r != ncclSuccess) what = "ncclGroupEnd";
            return first != ncclSuccess ? first : r;
        }), what);
        if (!posted) return;
        if (mine > 0) hk(hipMemcpyAsync(d_gather + offsets[0], eng.pcm_hbm(), (size_t)mine * 2, hipMemcpyDeviceToDevice, eng.stream), "own shard");
        if (gather_total > 0) hk(hipMemcpyAsync(h_gather, d_gather, (size_t)gather_total * 2, hipMemcpyDeviceToHost, eng.stream), "gather download");
        (void)wait_stream(sh, eng.stream, "gather");
    }

    void worker(int k) {
        Engine& eng = *engines[k];
        (void)hipSetDevice(eng.device);
        int64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_go.wait(lk, [&] { return stop || epoch != seen; });
                if (stop) return;
                seen = epoch;
            }
            Shard& sh = shards[k];
            const int nb = (int)sh.utt.size();
            if (nb > 0) {
                std::vector<const int32_t*> idp(nb); std::vector<int32_t> nn(nb), sd(nb); std::vector<float> l(nb);
                for (int i = 0; i < nb; i++) { const int u = sh.utt[i]; idp[i] = ids[u]; nn[i] = n[u]; sd[i] = sid ? sid[u] : 0; l[i] = ls ? ls[u] : 1.0f; }
                sh.rc = eng.run(nb, idp.data(), nn.data(), sd.data(), l.data());
                if (sh.rc == STS_OK && gather_mode == 1) {
                    sh.n_samples = eng.n_samples;           // the PCM stays on the device: rccl_gather() below
                } else if (sh.rc == STS_OK) {
                    sh.n_samples = eng.n_samples;
                    sh.pcm.resize((size_t)std::max<int64_t>(1, eng.total_samples));
                    if (eng.h_pcm) memcpy(sh.pcm.data(), eng.h_pcm, (size_t)eng.total_samples * 2);   // downloaded inside the run
                    else if (!eng.pcm_hbm() || hipMemcpyAsync(sh.pcm.data(), eng.pcm_hbm(), (size_t)eng.total_samples * 2, hipMemcpyDeviceToHost, eng.stream) != hipSuccess ||
                             hipStreamSynchronize(eng.stream) != hipSuccess) { sh.rc = STS_EDEVICE; sh.err = "PCM download failed"; }
                } else {
                    sh.err = eng.error();
                }
            }
            if (gather_mode == 1) {       // every rank takes part, also with an empty or failed shard
                {   // all shards have run before the first collective is posted: the bounded waits inside rccl_gather then cover the
                    // collective alone, not the slowest peer's synthesis (an imbalanced batch must not trip the timeout: ADVICE r04)
                    std::unique_lock<std::mutex> lk(mu);
                    const int64_t ge = gather_epoch;
                    if (++gather_arrived == (int)engines.size()) { gather_arrived = 0; gather_epoch++; cv_done.notify_all(); }
                    else cv_done.wait(lk, [&] { return gather_epoch != ge || stop; });
                }
                bool alive; { std::lock_guard<std::mutex> lk(mu); alive = !rccl_broken; }
                if (alive) rccl_gather(k);
                else if (sh.rc == STS_OK) { sh.rc = STS_EDEVICE; sh.err = "RCCL communicators were aborted by an earlier failure"; }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                pending--;
            }
            cv_done.notify_all();
        }
    }
};

static thread_local std::string g_multi_err;
static int multi_err(int code, const std::string& s) { g_multi_err = s; return code; }

// longest-first greedy bin packing by phoneme count; ties -> fewer utterances -> lower device index.  Deterministic.
static void shard_by_phonemes(int B, const int32_t* n, int ndev, std::vector<Shard>& out) {
    out.assign(ndev, Shard());
    std::vector<int> order(B);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return n[a] > n[b]; });
    std::vector<long> load(ndev, 0);
    for (int u : order) {
        int best = 0;
        for (int k = 1; k < ndev; k++)
            if (load[k] < load[best] || (load[k] == load[best] && out[k].utt.size() < out[best].utt.size())) best = k;
        out[best].utt.push_back(u);
        load[best] += n[u];
    }
    for (auto& s : out) std::sort(s.utt.begin(), s.utt.end());
}

extern "C" {

const char* sts_multi_last_error(void) { return g_multi_err.c_str(); }

// block of rank r starts at offsets[r] (in samples, 128-sample = 256-byte aligned so that every transfer starts on a cache-line
// pair); returns the total extent.  Host arithmetic only.
int64_t sts_multi_gather_layout(const int64_t* counts, int32_t n_ranks, int64_t* offsets) {
    int64_t off = 0;
    for (int r = 0; r < n_ranks; r++) {
        if (offsets) offsets[r] = off;
        const int64_t c = counts && counts[r] > 0 ? counts[r] : 0;
        off += (c + 127) / 128 * 128;
    }
    return off;
}

int sts_multi_gather_mode(const sts_multi* m) { return m ? m->gather_mode : 0; }

// the size of the handle's communicator as RCCL itself reports it (ncclCommCount on rank 0's communicator); 0: no RCCL gather on this
// handle (download mode, or the communicators were aborted), -1: the library does not export ncclCommCount
int sts_multi_rccl_ranks(sts_multi* m) {
    if (!m || m->gather_mode != 1 || m->comms.empty()) return 0;
    Rccl& R = rccl();
    if (!R.CommCount) return -1;
    int n = 0;
    const ncclResult_t r = m->rccl_call(0, [&](ncclComm_t cm) { return R.CommCount(cm, &n); });
    return r == ncclSuccess ? n : 0;
}
double sts_multi_last_gather_ms(const sts_multi* m) { return m ? m->last_gather_ms : 0.0; }
int sts_multi_set_conv_math(sts_multi* m, int mode) {
    if (!m) return multi_err(STS_EINVAL, "null handle");
    if (mode < 0 || mode > 3) return multi_err(STS_EINVAL, "conv math must be 0..3");
    for (auto& e : m->engines) { e->conv_math = mode; e->h2_disabled = false; e->h2_consecutive = 0; }
    return STS_OK;
}

// Test hook: the shared library that provides the nccl* entry points (default: librccl.so.1) and whether STS_MULTI_RCCL may list a
// device more than once (real RCCL refuses that; tests/fake_rccl emulates N ranks on ONE GPU).  Takes effect for handles created
// afterwards, and only before the first successful load in the process.
// Gated: refused unless the process environment carries STS_TEST_HOOKS=1 -- a caller of the drop-in library cannot swap the
// collective library or lift the distinct-device check by accident.
int sts_multi_set_rccl_library(const char* path, int allow_repeated_devices) {
    const char* gate = getenv("STS_TEST_HOOKS");
    if (!gate || strcmp(gate, "1") != 0) return multi_err(STS_ESTATE, "test hook: set STS_TEST_HOOKS=1 in the environment to enable it");
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    Rccl& R = rccl();
    if (R.h) return multi_err(STS_ESTATE, "an RCCL library is already loaded in this process");
    R.override_path = path ? path : "";
    R.allow_repeated = allow_repeated_devices != 0;
    return STS_OK;
}

int sts_multi_create(const float* blob, int64_t blob_bytes, const int32_t* devices, int32_t n_devices, sts_multi** out) {
    return sts_multi_create_ex(blob, blob_bytes, devices, n_devices, STS_MULTI_AUTO, out);
}

int sts_multi_create_ex(const float* blob, int64_t blob_bytes, const int32_t* devices, int32_t n_devices, int32_t flags, sts_multi** out) {
    if (!out) return multi_err(STS_EINVAL, "null out pointer");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return multi_err(STS_EINVAL, "device list must hold 1..64 entries");
    if (flags != STS_MULTI_AUTO && flags != STS_MULTI_RCCL && flags != STS_MULTI_DOWNLOAD) return multi_err(STS_EINVAL, "unknown flags");
    bool distinct = true;
    for (int a = 0; a < n_devices; a++) for (int b = a + 1; b < n_devices; b++) if (devices[a] == devices[b]) distinct = false;
    if (flags == STS_MULTI_RCCL && !distinct && !rccl().allow_repeated) return multi_err(STS_EINVAL, "the RCCL gather needs distinct devices (one communicator rank per GPU)");
    // the automatic mode = per-device downloads: the RCCL gather has never run on N > 1 real devices, so it is opt-in
    const bool want_rccl = flags == STS_MULTI_RCCL;
    sts_multi* m = new (std::nothrow) sts_multi();
    if (!m) return multi_err(STS_EDEVICE, "out of host memory");
    for (int k = 0; k < n_devices; k++) {
        m->engines.emplace_back(new Engine());
        m->engines.back()->polite_wait = true;      // worker threads sleep through most of a run's host waits (engine.hpp)
        m->engines.back()->host_pcm = !want_rccl;
        const int rc = m->engines.back()->init(blob, blob_bytes, devices[k]);
        if (rc != STS_OK) { multi_err(rc, "device " + std::to_string(devices[k]) + ": " + m->engines.back()->error()); delete m; return rc; }
    }
    m->shards.resize(n_devices);
    if (want_rccl) {
        std::lock_guard<std::mutex> lk(g_rccl_mu);
        Rccl& R = rccl();
        if (!R.load()) {
            if (flags == STS_MULTI_RCCL) { multi_err(STS_EDEVICE, R.err); delete m; return STS_EDEVICE; }
            for (auto& e : m->engines) e->host_pcm = true;                      // automatic mode: fall back to per-device downloads
        } else {
            m->comms.resize(n_devices);
            std::vector<int> devs(devices, devices + n_devices);
            const ncclResult_t r = R.CommInitAll(m->comms.data(), n_devices, devs.data());
            if (r != ncclSuccess) {
                m->comms.clear();
                if (flags == STS_MULTI_RCCL) { multi_err(STS_EDEVICE, std::string("ncclCommInitAll: ") + R.GetErrorString(r)); delete m; return STS_EDEVICE; }
                for (auto& e : m->engines) e->host_pcm = true;
            } else {
                m->gather_mode = 1;
                m->in_call.assign(n_devices, 0); m->abort_pending.assign(n_devices, 0);
                m->d_counts.assign(n_devices, nullptr);
                for (int k = 0; k < n_devices; k++) {
                    (void)hipSetDevice(devices[k]);
                    if (hipMalloc((void**)&m->d_counts[k], sizeof(long long) * (n_devices + 1)) != hipSuccess) { multi_err(STS_EDEVICE, "count buffer"); sts_multi_destroy(m); return STS_EDEVICE; }
                }
            }
        }
    }
    for (int k = 0; k < n_devices; k++) m->workers.emplace_back([m, k] { m->worker(k); });
    *out = m;
    return STS_OK;
}

void sts_multi_destroy(sts_multi* m) {
    if (!m) return;
    { std::lock_guard<std::mutex> lk(m->mu); m->stop = true; }
    m->cv_go.notify_all();
    for (auto& t : m->workers) t.join();
    for (size_t k = 0; k < m->comms.size(); k++) if (m->comms[k]) (void)rccl().CommDestroy(m->comms[k]);      // (aborted communicators were nulled)
    for (size_t k = 0; k < m->d_counts.size(); k++) if (m->d_counts[k]) { (void)hipSetDevice(m->engines[k]->device); (void)hipFree(m->d_counts[k]); }
    if (m->d_gather) { (void)hipSetDevice(m->engines[0]->device); (void)hipFree(m->d_gather); }
    if (m->h_gather) (void)hipHostFree(m->h_gather);
    delete m;
}

int sts_multi_device_count(const sts_multi* m) { return m ? (int)m->engines.size() : 0; }

int sts_multi_speaker_num(const sts_multi* m) {
    if (!m || m->engines.empty()) return 1;
    const Model& mm = m->engines[0]->model;
    return mm.spk_num == 0 ? 1 : mm.spk_num;
}

int sts_multi_shard_of(const sts_multi* m, int32_t B, const int32_t* n, int32_t* device_slot_out) {
    if (!m || !n || !device_slot_out || B <= 0) return multi_err(STS_EINVAL, "bad arguments");
    std::vector<Shard> sh;
    shard_by_phonemes(B, n, (int)m->engines.size(), sh);
    for (int k = 0; k < (int)sh.size(); k++) for (int u : sh[k].utt) device_slot_out[u] = k;
    return STS_OK;
}

int sts_multi_infer_ids_batch(sts_multi* m, int32_t B, const int32_t* const* ids, const int32_t* n, const int32_t* sid,
                              const float* length_scale, int16_t** pcm_out, int32_t* n_out) {
    if (!m || !ids || !n || !pcm_out || !n_out || B <= 0) return multi_err(STS_EINVAL, "bad arguments");
    for (int b = 0; b < B; b++) { pcm_out[b] = nullptr; n_out[b] = 0; }      // every output is defined before the first early return
    for (int b = 0; b < B; b++) if (n[b] <= 0 || !ids[b]) return multi_err(STS_EINVAL, "utterance with no phonemes");
    const int ndev = (int)m->engines.size();
    const bool was_gather = m->gather_mode == 1;
    {
        std::unique_lock<std::mutex> lk(m->mu);
        shard_by_phonemes(B, n, ndev, m->shards);
        m->B = B; m->ids = ids; m->n = n; m->sid = sid; m->ls = length_scale;
        m->pending = ndev;
        m->epoch++;
        m->cv_go.notify_all();
        m->cv_done.wait(lk, [&] { return m->pending == 0; });
    }
    int rc = STS_OK;
    const bool gathered = was_gather && !m->rccl_broken;
    if (was_gather && m->rccl_broken) {          // this call fails; the next ones bring the PCM home by per-device downloads
        m->gather_mode = 0;
        for (auto& e : m->engines) e->host_pcm = true;
        bool any = false;
        for (auto& sh : m->shards) any = any || sh.rc != STS_OK;
        if (!any) rc = multi_err(STS_EDEVICE, "RCCL gather failed (communicators aborted)");
    }
    for (int k = 0; k < ndev && rc == STS_OK; k++)
        if (m->shards[k].rc != STS_OK) rc = multi_err(m->shards[k].rc, "device slot " + std::to_string(k) + ": " + m->shards[k].err);
    if (rc == STS_OK && gathered) {
        // the counts every rank published must be the ones the engines reported on the host
        for (int k = 0; k < ndev && rc == STS_OK; k++) {
            int64_t want = 0;
            for (int32_t v : m->shards[k].n_samples) want += v;
            if (m->shards[k].utt.empty()) want = 0;
            if ((int)m->counts.size() != ndev || m->counts[k] != want) rc = multi_err(STS_EDEVICE, "RCCL gather: sample counts disagree");
        }
        if (rc == STS_OK && !m->h_gather && m->gather_total > 0) rc = multi_err(STS_EDEVICE, "RCCL gather: no host buffer");
    }
    for (int k = 0; k < ndev && rc == STS_OK; k++) {
        const Shard& sh = m->shards[k];
        size_t off = 0;
        const int16_t* src = gathered ? m->h_gather + m->offsets[k] : sh.pcm.data();
        for (size_t i = 0; i < sh.utt.size() && rc == STS_OK; i++) {
            const int u = sh.utt[i];
            const int32_t ns = sh.n_samples[i];
            pcm_out[u] = (int16_t*)malloc((size_t)(ns > 0 ? ns : 1) * 2);
            if (!pcm_out[u]) { rc = multi_err(STS_EDEVICE, "out of host memory"); break; }
            memcpy(pcm_out[u], src + off, (size_t)ns * 2);
            n_out[u] = ns;
            off += (size_t)ns;
        }
    }
    if (rc != STS_OK) for (int b = 0; b < B; b++) { free(pcm_out[b]); pcm_out[b] = nullptr; n_out[b] = 0; }
    return rc;
}

}  // extern "C"
