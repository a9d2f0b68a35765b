// multi.hip -- one host process, several MI355X: the multi-device form of sts_infer_ids_batch (SURVEY.md 8b "+ multi-device
// variant taking a device list", 8e).  No reference counterpart: SynthesizerTrn::infer handles one utterance on the CPU
// (/root/reference/src/models/SynthesizerTrn.cpp:323).
//
// Utterances are independent, so the batch is sharded BY UTTERANCE with no exchange step: weights are replicated (one
// engine = one stream set + ~116 MB of repacked weights per device), the shards are balanced longest-first by phoneme
// count (the same rule as summertts_amd/sharding.py: work is ~proportional to it), every device runs its shard as ONE
// packed variable-length batch on its own worker thread, downloads its int16 PCM through its own pinned staging buffer,
// and the caller gets the utterances back in INPUT order.  The result lives on the host (malloc'd per utterance, like
// sts_infer_ids_batch), so the "gather" is the per-device PCIe download -- there is nothing to move between GPUs; the
// RCCL gather of summertts_amd/sharding.py is for the one-process-per-GPU deployment, where rank 0 owns the output.
#include <stdlib.h>
#include <string.h>

#include <algorithm>
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
                if (sh.rc == STS_OK) {
                    sh.n_samples = eng.n_samples;
                    sh.pcm.resize((size_t)std::max<int64_t>(1, eng.total_samples));
                    if (eng.h_pcm) memcpy(sh.pcm.data(), eng.h_pcm, (size_t)eng.total_samples * 2);   // downloaded inside the run
                    else if (hipMemcpyAsync(sh.pcm.data(), eng.d_pcm, (size_t)eng.total_samples * 2, hipMemcpyDeviceToHost, eng.stream) != hipSuccess ||
                             hipStreamSynchronize(eng.stream) != hipSuccess) { sh.rc = STS_EDEVICE; sh.err = "PCM download failed"; }
                } else {
                    sh.err = eng.error();
                }
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

int sts_multi_create(const float* blob, int64_t blob_bytes, const int32_t* devices, int32_t n_devices, sts_multi** out) {
    if (!out) return multi_err(STS_EINVAL, "null out pointer");
    *out = nullptr;
    if (!devices || n_devices < 1 || n_devices > 64) return multi_err(STS_EINVAL, "device list must hold 1..64 entries");
    sts_multi* m = new (std::nothrow) sts_multi();
    if (!m) return multi_err(STS_EDEVICE, "out of host memory");
    for (int k = 0; k < n_devices; k++) {
        m->engines.emplace_back(new Engine());
        m->engines.back()->host_pcm = true;
        const int rc = m->engines.back()->init(blob, blob_bytes, devices[k]);
        if (rc != STS_OK) { multi_err(rc, "device " + std::to_string(devices[k]) + ": " + m->engines.back()->error()); delete m; return rc; }
    }
    m->shards.resize(n_devices);
    for (int k = 0; k < n_devices; k++) m->workers.emplace_back([m, k] { m->worker(k); });
    *out = m;
    return STS_OK;
}

void sts_multi_destroy(sts_multi* m) {
    if (!m) return;
    { std::lock_guard<std::mutex> lk(m->mu); m->stop = true; }
    m->cv_go.notify_all();
    for (auto& t : m->workers) t.join();
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
    for (int k = 0; k < ndev && rc == STS_OK; k++)
        if (m->shards[k].rc != STS_OK) rc = multi_err(m->shards[k].rc, "device slot " + std::to_string(k) + ": " + m->shards[k].err);
    for (int k = 0; k < ndev && rc == STS_OK; k++) {
        const Shard& sh = m->shards[k];
        size_t off = 0;
        for (size_t i = 0; i < sh.utt.size() && rc == STS_OK; i++) {
            const int u = sh.utt[i];
            const int32_t ns = sh.n_samples[i];
            pcm_out[u] = (int16_t*)malloc((size_t)(ns > 0 ? ns : 1) * 2);
            if (!pcm_out[u]) { rc = multi_err(STS_EDEVICE, "out of host memory"); break; }
            memcpy(pcm_out[u], sh.pcm.data() + off, (size_t)ns * 2);
            n_out[u] = ns;
            off += (size_t)ns;
        }
    }
    if (rc != STS_OK) for (int b = 0; b < B; b++) { free(pcm_out[b]); pcm_out[b] = nullptr; n_out[b] = 0; }
    return rc;
}

}  // extern "C"
