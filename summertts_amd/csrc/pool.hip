// pool.hip -- request scheduler over several engines on one GPU ("batch scheduler / packed
// varlen layout").  The reference synthesises one utterance per blocking call (SynthesizerTrn.cpp:323); a
// server in front of an MI355X wants (a) the latency-bound text side of one request overlapped with the
// decoder of another and (b) queued requests folded into one packed variable-length batch.  A pool owns
// N engines (one HIP stream set each, weights replicated: ~116 MB per engine out of 288 GB), one worker
// thread per engine, and one FIFO: a free worker takes the oldest request plus up to max_batch - 1 further
// queued ones, runs them as ONE Engine::run batch and completes their tickets.
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "engine.hpp"

using namespace sts;

namespace {
struct Request {
    int64_t ticket = 0;
    std::vector<int32_t> ids; int32_t sid = 0; float ls = 1.f;
    // result
    bool done = false, waited = false; int rc = STS_OK; std::string err;
    int16_t* pcm = nullptr; int32_t n = 0;
};
}  // namespace

struct sts_pool {
    std::vector<std::unique_ptr<Engine>> engines;
    std::vector<std::thread> workers;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::deque<std::shared_ptr<Request>> queue;
    std::map<int64_t, std::shared_ptr<Request>> pending;   // submitted, not yet collected
    int64_t next_ticket = 1;
    int max_batch = 8;
    bool stop = false;
    int64_t batches = 0, requests = 0;

    void worker(int k) {
        Engine& eng = *engines[k];
        (void)hipSetDevice(eng.device);
        for (;;) {
            std::vector<std::shared_ptr<Request>> take;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !queue.empty(); });
                if (stop && queue.empty()) return;
                while (!queue.empty() && (int)take.size() < max_batch) { take.push_back(queue.front()); queue.pop_front(); }
            }
            // run `grp` as one packed batch; on failure of a multi-request batch, re-run its members one by one so
            // that a bad request (e.g. an id outside the vocabulary) only fails itself
            auto run_group = [&](const std::vector<std::shared_ptr<Request>>& grp, auto&& self) -> void {
                const int B = (int)grp.size();
                std::vector<const int32_t*> idp(B); std::vector<int32_t> n(B), sid(B); std::vector<float> ls(B);
                for (int b = 0; b < B; b++) { idp[b] = grp[b]->ids.data(); n[b] = (int32_t)grp[b]->ids.size(); sid[b] = grp[b]->sid; ls[b] = grp[b]->ls; }
                int rc = eng.run(B, idp.data(), n.data(), sid.data(), ls.data());
                std::vector<int16_t> all;
                if (rc == STS_OK) {
                    all.resize((size_t)(eng.total_samples > 0 ? eng.total_samples : 1));
                    if (eng.h_pcm) memcpy(all.data(), eng.h_pcm, (size_t)eng.total_samples * 2);   // downloaded inside the run
                    else if (!eng.pcm_hbm() || hipMemcpyAsync(all.data(), eng.pcm_hbm(), (size_t)eng.total_samples * 2, hipMemcpyDeviceToHost, eng.stream) != hipSuccess ||
                             hipStreamSynchronize(eng.stream) != hipSuccess)
                        rc = STS_EDEVICE;
                }
                if (rc != STS_OK && B > 1) {
                    for (auto& r : grp) self(std::vector<std::shared_ptr<Request>>{r}, self);
                    return;
                }
                size_t off = 0;
                std::lock_guard<std::mutex> lk(mu);
                for (int b = 0; b < B; b++) {
                    Request& r = *grp[b];
                    r.rc = rc;
                    if (rc == STS_OK) {
                        r.n = eng.n_samples[b];
                        r.pcm = (int16_t*)malloc((size_t)(r.n > 0 ? r.n : 1) * 2);
                        if (r.pcm) memcpy(r.pcm, all.data() + off, (size_t)r.n * 2); else { r.rc = STS_EDEVICE; r.err = "out of host memory"; }
                        off += (size_t)r.n;
                    } else {
                        r.err = eng.error();
                    }
                    r.done = true;
                }
                batches++; requests += B;
            };
            run_group(take, run_group);
            cv_done.notify_all();
        }
    }
};

static thread_local std::string g_pool_err;
static int pool_err(int code, const std::string& s) { g_pool_err = s; return code; }

extern "C" {

const char* sts_pool_last_error(void) { return g_pool_err.c_str(); }

int sts_pool_create(const float* blob, int64_t blob_bytes, int device, int n_engines, int max_batch, sts_pool** out) {
    if (!out) return pool_err(STS_EINVAL, "null out pointer");
    *out = nullptr;
    if (n_engines < 1 || n_engines > 16 || max_batch < 1 || max_batch > 1024) return pool_err(STS_EINVAL, "n_engines in 1..16, max_batch in 1..1024");
    sts_pool* p = new (std::nothrow) sts_pool();
    if (!p) return pool_err(STS_EDEVICE, "out of host memory");
    p->max_batch = max_batch;
    for (int k = 0; k < n_engines; k++) {
        p->engines.emplace_back(new Engine());
        p->engines.back()->polite_wait = true;      // worker threads sleep through most of a run's host waits (engine.hpp)
        p->engines.back()->host_pcm = true;
        const int rc = p->engines.back()->init(blob, blob_bytes, device);
        if (rc != STS_OK) { pool_err(rc, p->engines.back()->error()); delete p; return rc; }
    }
    for (int k = 0; k < n_engines; k++) p->workers.emplace_back([p, k] { p->worker(k); });
    *out = p;
    return STS_OK;
}

void sts_pool_destroy(sts_pool* p) {
    if (!p) return;
    { std::lock_guard<std::mutex> lk(p->mu); p->stop = true; }
    p->cv_work.notify_all();
    for (auto& t : p->workers) t.join();
    for (auto& kv : p->pending) if (kv.second->pcm) free(kv.second->pcm);
    delete p;
}

int64_t sts_pool_submit(sts_pool* p, const int32_t* ids, int32_t n, int32_t sid, float length_scale) {
    if (!p || !ids || n <= 0) return pool_err(STS_EINVAL, "bad request");
    auto r = std::make_shared<Request>();
    r->ids.assign(ids, ids + n); r->sid = sid; r->ls = length_scale;
    {
        std::lock_guard<std::mutex> lk(p->mu);
        if (p->stop) return pool_err(STS_ESTATE, "pool is shutting down");
        r->ticket = p->next_ticket++;
        p->queue.push_back(r);
        p->pending[r->ticket] = r;
    }
    p->cv_work.notify_one();
    return r->ticket;
}

int sts_pool_wait(sts_pool* p, int64_t ticket, int16_t** pcm_out, int32_t* n_out) {
    if (!p || !pcm_out || !n_out) return pool_err(STS_EINVAL, "null argument");
    std::shared_ptr<Request> r;
    {
        std::unique_lock<std::mutex> lk(p->mu);
        auto it = p->pending.find(ticket);
        if (it == p->pending.end()) return pool_err(STS_EINVAL, "unknown ticket");
        r = it->second;
        if (r->waited) return pool_err(STS_EINVAL, "ticket is already being waited on");
        r->waited = true;
        p->cv_done.wait(lk, [&] { return r->done; });   // releases the mutex: `it` may be stale afterwards
        p->pending.erase(ticket);
    }
    if (r->rc != STS_OK) { if (r->pcm) free(r->pcm); return pool_err(r->rc, r->err); }
    *pcm_out = r->pcm; *n_out = r->n;
    return STS_OK;
}

int sts_pool_stats(sts_pool* p, int64_t* batches, int64_t* requests) {
    if (!p) return pool_err(STS_EINVAL, "null pool");
    std::lock_guard<std::mutex> lk(p->mu);
    if (batches) *batches = p->batches;
    if (requests) *requests = p->requests;
    return STS_OK;
}

}  // extern "C"
