// engine.hpp -- the batched acoustic pipeline on one MI355X: owns the HIP stream, the device-resident
// model, two workspace arenas (phoneme-level and frame-level) and a pinned staging buffer.
#pragma once
#include <hip/hip_runtime.h>

#include <deque>
#include <map>
#include <unordered_map>
#include <string>
#include <vector>

#include "../../include/summertts_hip.h"
#include "kernels.hpp"
#include "model.hpp"

namespace sts {

struct Arena {
    char* base = nullptr; size_t cap = 0, used = 0; bool measuring = false;
    void* alloc(size_t bytes) {
        size_t a = (used + 255) & ~(size_t)255;
        used = a + bytes;
        if (measuring) return nullptr;
        return base + a;
    }
    template <typename T> T* get(size_t count) { return (T*)alloc(count * sizeof(T)); }
};

// one packed level of the pipeline: which segments, how long, how many positions in total
struct Lvl {
    SegView seg; int nb = 0; int max_len = 0; long total = 0; long ld = 0;
};

struct ConvOpt {
    int in_act = 0; float slope = 0.f; int reflect = 0;
    int epi = EPI_STORE, epi_flag = 0; float epi_scale = 1.f;
    const float* res = nullptr; long res_ld = 0;
    float* aux = nullptr; long aux_ld = 0;
    int16_t* pcm = nullptr;
    const float* ubias = nullptr;
    int pad_l = -1;            // override the left padding (FFN same_padding)
    int tile = -1;             // force a kernel variant for this conv (as sts_set_conv_mode - 2); -1 = automatic
    int kslices = 1; long kslice_stride = 0;   // cross-workgroup split of K into partial outputs (kernels.hpp ConvArgs::kslices)
    // the conv's input is the mean ((x + sum1) [+ sum2]) / nsum of tensors of x's geometry (kernels.hpp ConvArgs::nsum); where the kernel
    // the conv is routed to cannot form it while staging, Engine::conv materialises it into sum_dst (sum_n floats) first
    const float* sum1 = nullptr; const float* sum2 = nullptr; int nsum = 0; float* sum_dst = nullptr; long sum_n = 0;
};

// streaming decode: PCM is handed to `cb` chunk by chunk (cb returns non-zero to stop)
struct StreamSpec { int chunk_frames; int (*cb)(void* user, const int16_t* pcm, int32_t n_samples, int32_t sample_offset); void* user; };
int decoder_halo_frames(const Model& M);

struct Tap { std::vector<float> data; int channels = 0; long length = 0; };

class Engine {
public:
    ~Engine();
    int init(const float* blob, int64_t bytes, int device);
    int run(int B, const int32_t* const* ids, const int32_t* n, const int32_t* sid, const float* ls, const StreamSpec* ss = nullptr);
    int run_once(int B, const int32_t* const* ids, const int32_t* n, const int32_t* sid, const float* ls, const StreamSpec* ss);
    const std::string& error() const { return err_; }

    Model model;
    int device = 0;
    // results of the last run
    std::vector<int32_t> n_samples;    // per utterance
    int64_t total_samples = 0;
    int16_t* d_pcm = nullptr;          // device, packed (the mapped pinned HOST buffer when pcm_in_host_: use pcm_hbm() where a true device buffer is required)
    // the last run's PCM in HBM, or null when no HBM copy exists (pcm_in_host_: the kernels wrote it over the host link) -- what RCCL
    // transfers and device-to-device copies must use (xGMI / D2D traffic out of host memory by accident)
    const int16_t* pcm_hbm() const { return pcm_in_host_ ? nullptr : d_pcm; }
    float host_us_setup_ = 0, host_us_enq_ = 0;
    double host_t0_ = 0, host_t_sync_ = 0;   // steady-clock microseconds: entry of the run, return of its last stream synchronisation
    bool pcm_in_host_ = false;         // this run's PCM was written by the kernels into the pinned host buffer (d_pcm is its device-side address; no HBM copy exists)
    const int16_t* h_pcm = nullptr;    // host copy of the last run's PCM (pinned, engine-owned) when host_pcm is set, else null
    bool host_pcm = false;             // download the PCM as part of run(): one stream sync per call instead of two
    std::vector<int32_t> durations_h;  // packed
    std::map<std::string, Tap> taps;
    sts_profile prof{};
    // controls
    std::vector<int32_t> forced_dur; bool have_forced = false;
    bool record_taps = false; int profiling = 0;       // profiling: 0 off, 1 all stage events, 2 the matrix-core region's two events only (sts_set_profiling)
    int conv_mode = 0;
    int conv_math = 3;                 // 0 = split-bf16 trunk convs (conv_bf3.hip), 1 = exact-fp32 MFMA, 3 = two-term fp16 (sts_set_conv_math)
    long h2_fallbacks = 0;             // runs repeated in split-bf16 because an activation left fp16's range (conv_math 3)
    int h2_consecutive = 0; bool h2_disabled = false;   // two repeats in a row: the engine stays on split-bf16 until sts_set_conv_math
    double products() const { return conv_math == 3 ? 3.0 : 6.0; }     // 16-bit matrix products per fp32 product
    int pcm_direct = 1;                // 1: with host_pcm, a one-utterance call's last kernel writes the PCM into the pinned host buffer itself (no download behind it) (STS_DBG_PCM_DIRECT)
    int dds_tail = 1;                  // 1: a ConvFlow's 29-row projection and spline step ride in its last DDSConv layer's launch (col_layer.hip tail) (STS_DBG_DDS_TAIL)
    int attn_reg = 1;                  // 1: one-query attention with its operands in registers (attention_reg_kernel) where the shape allows (STS_DBG_ATTN_REG)
    int attn_block_min_wgs = 96;       // attention_mfma_kernel from this many workgroups on (sts_debug_set)
    int launch_ahead = 1;              // 1: a one-utterance call the engine has served before enqueues flow + decoder before the frame count is on the host; 2 (tests): the memo is keyed by the phoneme count alone (sts_debug_set STS_DBG_LAUNCH_AHEAD)
    long ahead_misses = 0;             // launch-ahead calls whose count fell outside the predicted 64-frame bucket (a hash collision; repeated the waiting way)
    // the launch-ahead memo: hash of an utterance's (ids, speaker, length scale) -> its frame count (a pure function of them: the reference's
    // noise scale is 0); the last kMemoEntries distinct utterances, FIFO
    static constexpr size_t kMemoEntries = 1024;
    std::unordered_map<unsigned long long, long> seen_tf_; std::deque<unsigned long long> seen_order_;
    // Host waits of a run (the frame counts of a request the engine has not served before; the run's final stream synchronisation).  A direct
    // caller's thread polls them (lowest latency: it asked for one utterance now).  sts_pool / sts_multi workers set polite_wait: their thread
    // SLEEPS through ~80 % of the expected wait and polls only the tail, so N engines do not cost N host cores.  The
    // expectation is a running estimate per size class (log2 of the batch's phoneme count), scaled down for a smaller request of the class.
    bool polite_wait = false;
    struct WaitEst { double us = 0.0; long phonemes = 0; };
    WaitEst wait_est_[2][24];              // [0] frame counts, [1] final synchronisation
    void nap_before_wait(int which, long phonemes);
    void note_wait(int which, long phonemes, double waited_us, bool was_ready_at_wake);
    int final_sync(long phonemes);
    int ups_rowph = 1;                 // upsamplers: 1 the row-interleaved-phase copies (ConvArgs::rowph: whole-sector stores), 0 phase-major rows (sts_debug_set)
    int tail_fused = 1;                // MB-iSTFT / MS-iSTFT tail: 1 spectrum + inverse DFT / overlap-add + synthesis filter + int16 cast as one launch, 0 three (sts_debug_set)
    int chain_streams_dbg = -1;        // lab: stage mask -- the chains of the masked decoder stages as per-chain launches on three prioritised streams instead of grouped launches
    int h2p = 1;                       // 1: the wide ResBlock stages (C % 128 == 0) on pre-split channel-minor activations (conv_h2p.hip; two-term fp16 arithmetic only), 0: the staged kernels
    int h2p_tile = -1;                 // lab: tile code of conv_h2p_group (-1: automatic)
    int flow_fused = 1;                // 1: the reverse flow as one launch per WaveNet layer where eligible (wn_flow.hip; two-term fp16 arithmetic only);
                                       // 0: one launch per conv (sts_debug_set STS_DBG_FLOW_FUSED)
    hipStream_t stream = nullptr;

private:
    int fail(int code, const std::string& msg) { err_ = msg; return code; }
    bool ensure(Arena& a, size_t bytes);
    bool ensure_pinned(size_t bytes);
    ConvArgs conv_args(const DConv& c, const float* x, const Lvl& lin, float* y, const Lvl& lout, const ConvOpt& o, double* flops);
    void conv(const DConv& c, const float* x, const Lvl& lin, float* y, const Lvl& lout, const ConvOpt& o);
    void ln(const DLn& l, const float* a, const float* b, const float* res, float* y, const Lvl& lv, int pre_relu, int post_gelu, int nb = 1, long b_stride = 0);
    int pick_kslices(const DConv& c, const Lvl& lout) const;
    // (tail: the ConvFlow's projection + reverse spline step, run inside the last layer's launch when it can take them; *tail_done says so)
    struct SplineTail { const DConv* proj; float filter_sqrt; const float* r0; const float* r1; float* o0; float* o1; };
    float* dds(const DDds& d, float* h, float* t1, float* t2, const Lvl& lv, const DConv* pre = nullptr, const float* pre_in = nullptr,
               const float* pre_res = nullptr, const SplineTail* tail = nullptr, bool* tail_done = nullptr);
    struct RunCtx;                  // what the stages of one run share (engine.hip)
    int run_setup(RunCtx& c);
    int run_text_encoder(RunCtx& c);
    int run_durations(RunCtx& c);
    int run_frame_workspace(RunCtx& c);
    int run_flow(RunCtx& c);
    int run_decode(RunCtx& c, int nw, long Wtot, int maxW, int zoff0, int wlen0);
    int wait_frame_counts(RunCtx& c);
    int frame_geometry(RunCtx& c);
    int run_output(RunCtx& c);
    void tap(const char* name, const float* d, int channels, long ld, long length);
    void stage_begin(int s);
    void mark(int i);

    std::string err_;
    Arena arenaT_, arenaF_;
    char* pinned_ = nullptr; size_t pinned_cap_ = 0;
    char* pinned_pcm_ = nullptr; size_t pinned_pcm_cap_ = 0;
    int16_t* pinned_pcm_dev_ = nullptr;   // the same buffer as the GPU sees it (mapped): the last kernel of a small call writes the PCM straight into it
    unsigned* ovf_host_ = nullptr; unsigned* ovf_ = nullptr;              // conv_math 3: overflow word (host-mapped) and its device address
    int* hmap_ = nullptr; int* hmap_dev_ = nullptr; size_t hmap_cap_ = 0;   // host-mapped result block of the durations kernel
    unsigned* arrive_ = nullptr; int seq_ = 0;
    hipEvent_t ev_[8] = {};
    static constexpr int kAux = 3;            // ResBlock chains of one decoder stage run concurrently
    hipStream_t aux_[kAux] = {};
    hipEvent_t ev_fork_ = nullptr, ev_join_[kAux] = {};
    hipEvent_t ev_setup_ = nullptr;      // behind run_setup's upload of the pinned staging block (batches)
    hipStream_t cur_ = nullptr;               // stream the conv()/ln() helpers launch on
    bool have_events_ = false;
    int cur_stage_ = 0;
    double flops_[4] = {0, 0, 0, 0};
    double bytes_[4] = {0, 0, 0, 0};
    double bytes_w_[4] = {0, 0, 0, 0};       // the weight share of bytes_ (does not scale with the positions)
    double mfma_flops_ = 0, mfma_exec_ = 0, bf16_exec_ = 0, sync_wait_ms_ = 0; int mfma_launches_ = 0; bool in_mfma_region_ = false;
};

}  // namespace sts
