// engine.hip -- ids -> int16 PCM on one MI355X.  Restates the pipeline of
// /root/reference/src/models/SynthesizerTrn.cpp:357-400 for a BATCH of utterances packed along time:
//   TextEncoder -> duration predictor -> [one D2H of the frame counts] -> length regulator ->
//   reverse flow -> decoder -> int16.
// Everything runs on the engine's own HIP stream; the only host synchronisation inside a run is the
// data-dependent frame count F (SynthesizerTrn.cpp:376-381).
#include "engine.hpp"
#include <time.h>
#include "knobs.hpp"
#include "../../include/tts_logger.h"

#include <algorithm>
#include <chrono>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <thread>

namespace sts {

#define HIPCK(call)                                                                           \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) return fail(STS_EDEVICE, std::string(#call) + ": " + hipGetErrorString(e__)); \
    } while (0)

Engine::~Engine() {
    if (stream) (void)hipStreamSynchronize(stream);
    free_model(model);
    if (arenaT_.base) (void)hipFree(arenaT_.base);
    if (arenaF_.base) (void)hipFree(arenaF_.base);
    if (pinned_) (void)hipHostFree(pinned_);
    if (pinned_pcm_) (void)hipHostFree(pinned_pcm_);
    if (ovf_host_) (void)hipHostFree(ovf_host_);
    if (hmap_) (void)hipHostFree(hmap_);
    if (arrive_) (void)hipFree(arrive_);
    if (have_events_) {
        for (auto& e : ev_) (void)hipEventDestroy(e);
        (void)hipEventDestroy(ev_fork_);
        if (ev_setup_) (void)hipEventDestroy(ev_setup_);
        for (auto& e : ev_join_) (void)hipEventDestroy(e);
        for (auto& a : aux_) if (a) (void)hipStreamDestroy(a);
    }
    if (stream) (void)hipStreamDestroy(stream);
}

static int default_conv_math();
static constexpr int MAX_HALO_H2P = 64;      // conv_common.hpp MAX_HALO: the staged window of conv_h2p.hip
// one polite spin-wait step on whatever host this is built for (the x86 builtin alone broke aarch64 / ppc64 builds)
static inline void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__) || defined(__arm__)
    asm volatile("yield" ::: "memory");
#else
    std::this_thread::yield();
#endif
}

int Engine::init(const float* blob, int64_t bytes, int dev) {
    conv_math = default_conv_math();
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0)
        return fail(STS_EDEVICE, "no HIP device visible: the SummerTTS HIP engine needs an AMD GPU (gfx950) -- there is no CPU fallback");
    if (dev < 0 || dev >= count) return fail(STS_EINVAL, "device index out of range");
    device = dev;
    HIPCK(hipSetDevice(dev));
    HIPCK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    for (auto& e : ev_) HIPCK(hipEventCreate(&e));
    HIPCK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
    HIPCK(hipEventCreateWithFlags(&ev_setup_, hipEventDisableTiming));
    for (auto& e : ev_join_) HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    {   // the chain with the largest kernel size (most FLOPs) is the critical path of a decoder stage:
        // give it the highest queue priority, the lightest chain the lowest
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // lo = least priority (numerically greatest)
        for (int k = 0; k < kAux; k++) {
            int pr = lo + (hi - lo) * k / (kAux > 1 ? kAux - 1 : 1);
            HIPCK(hipStreamCreateWithPriority(&aux_[k], hipStreamNonBlocking, pr));
        }
    }
    cur_ = stream;
    have_events_ = true;
    if (!blob || bytes < 32) return fail(STS_EMODEL, "model blob too small");
    if (!load_model(blob, bytes / (int64_t)sizeof(float), model)) return fail(STS_EMODEL, "model parse failed: " + model.error);
    return STS_OK;
}

bool Engine::ensure(Arena& a, size_t bytes) {
    if (bytes <= a.cap) return true;
    (void)hipStreamSynchronize(stream);
    if (a.base) (void)hipFree(a.base);
    a.base = nullptr; a.cap = 0;
    size_t want = bytes + bytes / 4 + (1u << 20);
    if (hipMalloc((void**)&a.base, want) != hipSuccess) { if (hipMalloc((void**)&a.base, bytes) != hipSuccess) return false; want = bytes; }
    a.cap = want;
    return true;
}

bool Engine::ensure_pinned(size_t bytes) {
    if (bytes <= pinned_cap_) return true;
    (void)hipStreamSynchronize(stream);
    if (pinned_) (void)hipHostFree(pinned_);
    pinned_ = nullptr; pinned_cap_ = 0;
    size_t want = bytes * 2 + 4096;
    if (hipHostMalloc((void**)&pinned_, want, hipHostMallocDefault) != hipSuccess) return false;
    pinned_cap_ = want;
    return true;
}

void Engine::stage_begin(int s) { cur_stage_ = s; }
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// profiling 1: all eight stage events; 2: only the two that bracket the decoder's matrix-core region (events 5 / 6) -- every event is a barrier
// packet between two kernels (5-10 us of bubble each, docs/HISTORY.md 12-4), so the headline step pays for two of them, not eight
void Engine::mark(int i) { if (profiling == 1 || (profiling == 2 && (i == 5 || i == 6))) (void)hipEventRecord(ev_[i], stream); }

// builds the kernel arguments of one conv and books its FLOPs / minimum HBM bytes
// default arithmetic of the trunk convs: STS_CONV_MATH = f16x2 (two fp16 terms, three products: the default) | bf16x3 (three bf16
// terms, six products) | f32 (the exact-fp32 MFMA instruction)          (conv_bf3.hip)
static int default_conv_math() {
    const char* v = getenv("STS_CONV_MATH");
    if (!v || !*v) return 3;
    if (!strcmp(v, "f16x2") || !strcmp(v, "3")) return 3;
    if (!strcmp(v, "bf16x3") || !strcmp(v, "0")) return 0;
    if (!strcmp(v, "f32") || !strcmp(v, "fp32") || !strcmp(v, "1")) return 1;
    // (a typo such as "bf16" used to select a default silently)
    tts_log(TTS_LOG_WARNING, (std::string("summertts_hip: STS_CONV_MATH=") + v + " is not one of f16x2 | bf16x3 | f32 -- ignored, the default (f16x2) applies\n").c_str());
    return 3;
}

ConvArgs Engine::conv_args(const DConv& c, const float* x, const Lvl& lin, float* y, const Lvl& lout, const ConvOpt& o, double* flops) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.x_ld = lin.ld; a.y = y; a.y_ld = lout.ld;
    a.w = c.w; a.wb3 = c.wb3; a.bias = c.bias; a.ubias = o.ubias; a.ubias_ld = lout.nb;
    if (conv_math == 3 && c.wh2) { a.wb3 = c.wh2; a.math = 1; a.wscale = c.h2_scale; a.ovf = ovf_; }     // "f16x2": the two-term fp16 copy
    a.res = o.res; a.res_ld = o.res_ld ? o.res_ld : lout.ld;
    a.aux = o.aux; a.aux_ld = o.aux_ld ? o.aux_ld : lout.ld;
    a.pcm = o.pcm;
    a.Cin = c.Cin; a.Cout = c.Cout; a.Cin_pad = c.Cin_pad; a.Cout_pad = c.Cout_pad;
    if (c.transposed) {
        a.ntap = c.J; a.tap_step = -1; a.tap_off = 0; a.out_stride = c.stride; a.out_off = -c.pad;
        a.transposed = 1; a.n_extra = c.J - 1; a.max_n = lin.max_len + c.J - 1;
    } else {
        a.ntap = c.k; a.tap_step = c.dil; a.tap_off = -(o.pad_l >= 0 ? o.pad_l : c.pad); a.out_stride = 1; a.out_off = 0;
        a.max_n = lout.max_len;
    }
    a.depthwise = c.depthwise;
    a.in_act = o.in_act; a.in_slope = o.slope; a.in_reflect = o.reflect;
    a.epi = o.epi; a.epi_flag = o.epi_flag; a.epi_scale = o.epi_scale; a.H = c.H; a.gate_perm = c.gate_perm;
    a.in_seg = lin.seg; a.out_seg = lout.seg; a.B = lout.nb;
    a.kslices = o.kslices; a.kslice_stride = o.kslice_stride;
    if (o.nsum >= 2) { a.xs1 = o.sum1; a.xs2 = o.sum2; a.nsum = o.nsum; }
    const double positions = c.transposed ? (double)lin.total : (double)lout.total;
    const double fl = 2.0 * c.macs_per_out * positions;
    flops_[cur_stage_] += fl;
    {   // algorithmic HBM bytes: input once, output once, weights once (+ residual / read-modify-write operands)
        const size_t wfl = c.depthwise ? (size_t)c.k * c.Cout : (size_t)c.k * c.Cin * c.Cout;
        double by = 4.0 * ((double)c.Cin * lin.total + (double)c.Cout * lout.total);
        if (o.res || o.epi == EPI_SUB || o.epi == EPI_RESSKIP) by += 4.0 * (double)c.Cout * lout.total;
        bytes_[cur_stage_] += by;                         // (scales with the positions)
        bytes_w_[cur_stage_] += 4.0 * (double)wfl;        // (does not)
    }
    if (flops) *flops = fl;
    return a;
}

void Engine::conv(const DConv& c, const float* x, const Lvl& lin, float* y, const Lvl& lout, const ConvOpt& o) {
    double fl = 0;
    ConvArgs a = conv_args(c, x, lin, y, lout, o, &fl);
    const bool can_mfma = conv_mode != 1 && conv_mfma_eligible(a);
    // split-bf16 arithmetic: the decoder trunk always; other matrix-core convs (flow, text encoder, conv_pre) from the grid size on
    // at which they stop being launch-latency-bound (a batch of a few dozen utterances); conv_math 2 = wherever eligible (tests)
    const bool use_bf3 = conv_mode == 0 && conv_math != 1 && o.tile < 0 && conv_bf3_eligible(a) &&
                         (in_mfma_region_ || conv_math == 2 || conv_bf3_blocks(a) >= 384);
    if (use_bf3 && c.transposed && ups_rowph && a.epi == EPI_STORE && !a.ubias && (a.math == 1 ? c.wh2r : c.wb3r)) {
        // upsamplers: the copy whose rows interleave the phases (rho = cout * stride + phase) -- same sums in the same order, whole-sector stores
        a.wb3 = a.math == 1 ? c.wh2r : c.wb3r;
        a.bias = c.bias ? c.bias_r : nullptr;
        a.Cout = a.Cout_pad = c.Cout_pad * c.stride;
        a.rowph = c.stride;
    }
    if (a.nsum >= 2 && !(use_bf3 ? conv_bf3_takes_sum(a) : (!can_mfma && conv_cout1_takes(a)))) {
        // this conv's kernel cannot form the mean of its input terms while staging: one launch materialises it (the round-3 form)
        const float* r[3] = {x, o.sum1, o.sum2};
        sum_scale(o.sum_dst, r, a.nsum, o.sum_n, cur_);
        a.x = o.sum_dst; a.xs1 = a.xs2 = nullptr; a.nsum = 0;
    }
    if (use_bf3) {
        if (in_mfma_region_) { mfma_flops_ += fl; bf16_exec_ += products() * fl; mfma_launches_++; }
        static const int bt = exp_int("STS_BF3_TILE", -1);   // experiment knob
        conv_bf3(a, cur_, bt);
    } else if (can_mfma) {
        if (in_mfma_region_) { mfma_flops_ += fl; mfma_exec_ += fl; mfma_launches_++; }
        conv_mfma(a, cur_, o.tile >= 0 ? o.tile : (conv_mode >= 2 ? conv_mode - 2 : -1));
    } else {
        // the generic kernel knows nothing of cross-workgroup K slices: it writes the complete sum into slice 0, so the
        // consumer's other partial operands must read as zero
        if (a.kslices > 1 && a.kslice_stride > 0)
            (void)hipMemsetAsync(a.y + a.kslice_stride, 0, (size_t)(a.kslices - 1) * (size_t)a.kslice_stride * sizeof(float), cur_);
        conv_generic(a, cur_);
    }
}

// Cross-workgroup K split for a conv whose output has too few tiles to fill the chip while its K loop is long (text-encoder
// FFN second conv at batch 1: 24 tiles x K = 2304): slices are sized so that about one workgroup per CU results and
// every slice keeps >= 24 groups of 8 input channels.  1 = no split.
int Engine::pick_kslices(const DConv& c, const Lvl& lout) const {
    if (conv_mode != 0 || c.depthwise || c.transposed || c.Cin < 32) return 1;
    const long tiles = (long)((lout.max_len + 31) / 32) * (c.Cout_pad / 32) * lout.nb;
    const long groups = (long)c.k * (c.Cin_pad / 8);
    int s = 1;
    while (s < 8 && tiles * (s * 2) <= 256 && groups / (s * 2) >= 24) s *= 2;
    return s;
}

void Engine::ln(const DLn& l, const float* a, const float* b, const float* res, float* y, const Lvl& lv, int pre_relu, int post_gelu, int nb, long b_stride) {
    LnArgs g;
    memset(&g, 0, sizeof(g));
    g.a = a; g.a_ld = lv.ld; g.b = b; g.b_ld = lv.ld; g.res = res; g.res_ld = lv.ld; g.y = y; g.y_ld = lv.ld;
    g.nb = nb; g.b_stride = b_stride;
    g.gamma = l.g; g.beta = l.b; g.C = l.C; g.pre_relu = pre_relu; g.post_gelu = post_gelu;
    g.seg = lv.seg; g.B = lv.nb; g.max_len = lv.max_len;
    bytes_[cur_stage_] += 4.0 * (double)l.C * (double)lv.total * (2.0 + (b ? 1.0 : 0.0) + (res ? 1.0 : 0.0));
    layer_norm(g, cur_);
}

// /root/reference/src/modules/DDSConv.cpp:84-111: x += gelu(LN2(conv1x1(gelu(LN1(dwconv(x))))))
// Returns the buffer that holds the result (h or t1: the fused layers ping-pong between them).
// `pre` (optional): the ConvFlow's 1 -> C input conv, h = (pre(pre_in) + pre_res) (/root/reference/src/modules/ConvFlow.cpp:252-254;
// pre_in == null: the all-zero latent).  Where the first layer runs fused, h is never materialised: the kernel evaluates it at its
// depthwise taps; otherwise the conv runs first, into `h`.
float* Engine::dds(const DDds& d, float* h, float* t1, float* t2, const Lvl& lv, const DConv* pre, const float* pre_in, const float* pre_res,
                   const SplineTail* tail, bool* tail_done) {
    if (tail_done) *tail_done = false;
    static const bool no_col = exp_flag("STS_NO_COL_LAYER");   // experiment knob
    float* cur = h;
    bool pre_pending = pre != nullptr;
    auto run_pre = [&]() {      // the unfused form of the input conv (zero input: a memset feeds it)
        if (!pre_pending) return;
        pre_pending = false;
        const float* in = pre_in;
        if (!in) { (void)hipMemsetAsync(t2, 0, (size_t)lv.total * sizeof(float), cur_); in = t2; }
        ConvOpt oa; oa.epi = EPI_RESADD; oa.res = pre_res;
        conv(*pre, in, lv, h, lv, oa);
    };
    for (int i = 0; i < d.n; i++) {
        const DConv& c = d.sep[i];
        const DConv& pw = d.pw[i];
        if (!no_col && conv_mode != 1 && !pw.depthwise && pw.k == 1 && pw.Cin == pw.Cout && pw.Cin == d.n1[i].C && pw.Cin == d.n2[i].C &&
            pw.Cin_pad == pw.Cin) {
            // the whole layer as ONE launch (col_layer.hip); output goes to the other buffer: neighbouring workgroups still
            // read `cur` through the dilated depthwise taps
            ColLayerArgs g;
            memset(&g, 0, sizeof(g));
            float* nxt = cur == h ? t1 : h;
            g.x = cur; g.x_ld = lv.ld;
            g.dw_w = c.w; g.dw_b = c.bias; g.dw_k = c.k; g.dw_dil = c.dil; g.dw_pad = c.pad; g.dw_ld = c.Cout_pad;
            g.g1 = d.n1[i].g; g.b1 = d.n1[i].b;
            g.wc = pw.wc; g.bias = pw.bias;
            g.g2 = d.n2[i].g; g.b2 = d.n2[i].b; g.post_gelu = 1;
            g.res = cur; g.res_ld = lv.ld; g.y = nxt; g.y_ld = lv.ld; g.C = pw.Cin;
            g.seg = lv.seg; g.B = lv.nb; g.max_len = lv.max_len;
            if (pre_pending && i == 0 && pre->Cin == 1 && pre->k == 1 && pre->Cout == pw.Cin && !pre->depthwise && pre_res) {
                // fold h = (w r + b) + g into the layer: x = res = g, rank-1 term from the conv's single input row
                g.x = pre_res; g.res = pre_res; g.xs_w = pre->w; g.xs_b = pre->bias; g.xr = pre_in;
                if (col_layer_eligible(g)) {
                    pre_pending = false;
                    flops_[cur_stage_] += 2.0 * pre->macs_per_out * (double)lv.total;
                } else { g.x = cur; g.res = cur; g.xs_w = g.xs_b = g.xr = nullptr; }
            }
            run_pre();
            if (tail && tail_done && dds_tail && i == d.n - 1 && tail->proj && tail->proj->k == 1 && !tail->proj->depthwise && tail->proj->Cout == 29 &&
                tail->proj->Cout_pad == 32 && tail->proj->Cin == pw.Cin && tail->proj->Cin_pad == pw.Cin && tail->proj->w && tail->proj->bias) {
                // the layer's output feeds only the projection: both and the spline step in this launch
                ColLayerArgs gt = g;
                gt.tp_w = tail->proj->w; gt.tp_b = tail->proj->bias; gt.tp_fs = tail->filter_sqrt;
                gt.tp_r0 = tail->r0; gt.tp_r1 = tail->r1; gt.tp_o0 = tail->o0; gt.tp_o1 = tail->o1;
                if (col_layer_eligible(gt)) {
                    flops_[cur_stage_] += 2.0 * (c.macs_per_out + pw.macs_per_out + tail->proj->macs_per_out) * (double)lv.total;
                    bytes_[cur_stage_] += 4.0 * ((double)pw.Cin * lv.total + (double)pw.Cin * pw.Cout + (double)pw.Cin * 32 + 4.0 * lv.total);
                    col_layer(gt, cur_);
                    *tail_done = true;
                    return nxt;                 // (not written: the caller ignores it when the tail ran)
                }
            }
            if (col_layer_eligible(g)) {
                flops_[cur_stage_] += 2.0 * (c.macs_per_out + pw.macs_per_out) * (double)lv.total;
                bytes_[cur_stage_] += 4.0 * ((double)pw.Cin * lv.total * 2.0 + (double)pw.Cin * pw.Cout);
                col_layer(g, cur_);
                cur = nxt;
                continue;
            }
        }
        run_pre();
        {   // depthwise conv fused into the LayerNorm that consumes it (one launch instead of two)
            LnArgs g;
            memset(&g, 0, sizeof(g));
            g.a = cur; g.a_ld = lv.ld; g.y = t2; g.y_ld = lv.ld;
            g.gamma = d.n1[i].g; g.beta = d.n1[i].b; g.C = d.n1[i].C; g.post_gelu = 1;
            g.dw_w = c.w; g.dw_b = c.bias; g.dw_k = c.k; g.dw_dil = c.dil; g.dw_pad = c.pad; g.dw_ld = c.Cout_pad;
            g.seg = lv.seg; g.B = lv.nb; g.max_len = lv.max_len;
            flops_[cur_stage_] += 2.0 * c.macs_per_out * (double)lv.total;
            bytes_[cur_stage_] += 4.0 * (double)g.C * (double)lv.total * 2.0;
            layer_norm(g, cur_);
        }
        float* other = cur == h ? t1 : h;       // conv output; `cur` stays the residual and is updated in place
        conv(pw, t2, lv, other, lv, ConvOpt());
        ln(d.n2[i], other, nullptr, cur, cur, lv, 0, 1);
    }
    return cur;
}


void Engine::tap(const char* name, const float* d, int channels, long ld, long length) {
    if (!record_taps) return;
    Tap& t = taps[name];
    t.channels = channels; t.length = length;
    t.data.resize((size_t)channels * length);
    (void)hipStreamSynchronize(stream);
    (void)hipMemcpy2D(t.data.data(), (size_t)length * sizeof(float), d, (size_t)ld * sizeof(float),
                      (size_t)length * sizeof(float), (size_t)channels, hipMemcpyDeviceToHost);
}


// Frames of z (per side) that influence one output sample through the decoder: conv_pre, every upsampler,
// the widest ResBlock chain of every stage, and the tail.  Conservative (rounded up at every level).
int decoder_halo_frames(const Model& M) {
    double R = M.dec_type == 0 ? (M.conv_post.k - 1) / 2 : (M.conv_post.k - 1) / 2 + 12;   // tail: reflect pad, iSTFT overlap, synthesis FIR
    const int nk = M.n_resk;
    for (int i = M.n_up - 1; i >= 0; i--) {
        double rb = 0;
        for (int j = 0; j < nk; j++) {
            const DResBlock& b = M.rb[(size_t)i * nk + j];
            double r = 0;
            for (size_t d = 0; d < b.c1.size(); d++) r += b.c1[d].dil * (b.c1[d].k - 1) / 2 + b.c2[d].dil * (b.c2[d].k - 1) / 2;
            if (r > rb) rb = r;
        }
        R += rb;
        const DConv& up = M.ups[i];
        R = ceil((R + up.k) / (double)up.stride) + 1;     // through the transposed conv (kernel k, stride s)
    }
    R += (M.conv_pre.k - 1) / 2;
    return (int)ceil(R) + 1;
}

struct BufT {
    int *meta_i; float* ls; int* ids; int* forced;
    float *x, *qkv, *att, *y, *x1, *ffh, *m;
    float *dh, *dt1, *dt2, *dc, *dhh, *dp29, *dr[4], *dlogw;
    int *dur, *cum, *frames;
    float *g, *cond_dp, *cond_dec, *cond_wn;
};
struct BufF {
    float *z, *h, *acts, *out, *x0, *regA, *regB, *tailA, *tailB, *tailC, *wave, *fliptmp;
    float *ff_h[2], *ff_part[2], *ff_macc[2], *ff_alt;        // one-launch-per-layer flow (wn_flow.hip): channel-minor h / partial sums / -m slices, alternate home of a z half
    int16_t* pcm;
};

// Everything a run's stages share: batch geometry, workspace pointers, host / device tables.  Engine::run() fills it stage by
// stage; the stage functions below see its fields under the names the pipeline has always used (RUN_ALIASES).
struct Engine::RunCtx {
    int B = 0; const int32_t* const* ids = nullptr; const int32_t* n = nullptr; const int32_t* sid = nullptr; const float* ls = nullptr;
    const StreamSpec* ss = nullptr;
    std::vector<int> offT, lenT; long Ttot = 0; int maxT = 0;
    int H = 0, C = 0, FF = 0, fdp = 0, wnH = 0, wnL = 0, ffn2_slices = 1;
    Lvl lvT, lvB, lv1;
    BufT bt; BufF bf;
    size_t meta_ints = 0, up_bytes = 0;
    int *pm = nullptr, *p_offT = nullptr, *p_lenT = nullptr, *p_sid = nullptr, *p_offF = nullptr, *p_lenF = nullptr, *p_one = nullptr;
    int *d_offT = nullptr, *d_lenT = nullptr, *d_sid = nullptr, *d_offF = nullptr, *d_lenF = nullptr, *d_one = nullptr, *d_win = nullptr;
    bool inl = false, no_inline_seg = false, ms = false;
    long Ftot = 0; int maxF = 0, hop = 0;
    // One utterance (the reference's own call shape): buffers, leading dimensions and every dispatch decision use the frame CAPACITY
    // Fld = the count rounded up to a bucket of 64 frames, so that a call which launches the flow and the decoder AHEAD of the frame count
    // (ahead: the count is predicted, the kernels read the real one from device memory) makes exactly the dispatch decisions of a call
    // that waited for it -- and returns bit-identical samples.  Batches: Fld == Ftot, nothing changes.
    long Fld = 0; int maxFld = 0; bool ahead = false, ahead_b = false, mapped = false, forced = false;
    std::vector<unsigned long long> req_keys; std::vector<long> predF;      // launch-ahead memo: per-utterance request hashes, remembered frame counts (empty: not all known)
    int halo = 0; long Wcap = 0; int upS = 1; long Lsb = 0; int sbC = 0;
    bool use_ff = false; int ffG = 0;
};
#define RUN_ALIASES(c)                                                                                                              \
    [[maybe_unused]] Model& M = model;                                                                                              \
    [[maybe_unused]] const int B = (c).B; [[maybe_unused]] const StreamSpec* const ss = (c).ss;                                     \
    [[maybe_unused]] const long Ttot = (c).Ttot; [[maybe_unused]] const int maxT = (c).maxT;                                        \
    [[maybe_unused]] const int H = (c).H, C = (c).C, FF = (c).FF, fdp = (c).fdp, wnH = (c).wnH, wnL = (c).wnL, ffn2_slices = (c).ffn2_slices; \
    [[maybe_unused]] Lvl &lvT = (c).lvT, &lvB = (c).lvB, &lv1 = (c).lv1;                                                            \
    [[maybe_unused]] BufT& bt = (c).bt; [[maybe_unused]] BufF& bf = (c).bf;                                                         \
    [[maybe_unused]] const size_t up_bytes = (c).up_bytes;                                                                          \
    [[maybe_unused]] int* const pm = (c).pm; [[maybe_unused]] int* const p_offF = (c).p_offF; [[maybe_unused]] int* const p_lenF = (c).p_lenF; \
    [[maybe_unused]] int* const d_sid = (c).d_sid; [[maybe_unused]] int* const d_offF = (c).d_offF; [[maybe_unused]] int* const d_lenF = (c).d_lenF; \
    [[maybe_unused]] int* const d_win = (c).d_win;                                                                                  \
    [[maybe_unused]] const bool inl = (c).inl, no_inline_seg = (c).no_inline_seg, ms = (c).ms;                                      \
    [[maybe_unused]] const long Ftot = (c).Ftot; [[maybe_unused]] const int maxF = (c).maxF, hop = (c).hop;                         \
    [[maybe_unused]] const long Fld = (c).Fld; [[maybe_unused]] const int maxFld = (c).maxFld; [[maybe_unused]] const bool ahead = (c).ahead; \
    [[maybe_unused]] const int halo = (c).halo; [[maybe_unused]] const long Wcap = (c).Wcap, Lsb = (c).Lsb;                         \
    [[maybe_unused]] const int upS = (c).upS, sbC = (c).sbC;                                                                        \
    [[maybe_unused]] const bool use_ff = (c).use_ff; [[maybe_unused]] const int ffG = (c).ffG;

// ---- stage 0: batch geometry at the phoneme level, phoneme-level workspace, the one host-to-device copy of a run
int Engine::run_setup(RunCtx& c) {
    Model& M = model;
    const int B = c.B; const int32_t* const* ids = c.ids; const int32_t* n = c.n; const int32_t* sid = c.sid; const float* ls = c.ls;
    // ---------------- host-side batch geometry (phoneme level)
    std::vector<int>& offT = c.offT; std::vector<int>& lenT = c.lenT;
    offT.assign(B, 0); lenT.assign(B, 0);
    long& Ttot = c.Ttot; int& maxT = c.maxT;
    Ttot = 0; maxT = 0;
    for (int b = 0; b < B; b++) {
        if (n[b] <= 0 || !ids[b]) return fail(STS_EINVAL, "utterance with no phonemes");
        for (int i = 0; i < n[b]; i++)
            if (ids[b][i] < 0 || ids[b][i] >= M.vocab) return fail(STS_EINVAL, "phoneme id outside the model vocabulary");
        offT[b] = (int)Ttot; lenT[b] = n[b]; Ttot += n[b]; if (n[b] > maxT) maxT = n[b];
    }
    if (Ttot > (1 << 24)) return fail(STS_EINVAL, "batch too large");
    if ((size_t)(M.hidden / 2 + 8 + 5 * (size_t)maxT) * 4 > 150 * 1024) return fail(STS_EINVAL, "utterance too long for the attention kernel");
    if (have_forced && (long)forced_dur.size() != Ttot) { have_forced = false; return fail(STS_EINVAL, "forced durations do not match the batch"); }

    const int H = c.H = M.hidden, C = c.C = M.inter;
    const int FF = c.FF = M.n_layers ? M.ffn[0].c1.Cout : 0;
    const int fdp = c.fdp = M.dur_type == 0 ? M.sdp_pre.Cout : M.fix_c1.Cout;
    const int wnH = c.wnH = M.n_flows ? M.cp[0].wn.H : 0;
    const int wnL = c.wnL = M.n_flows ? M.cp[0].wn.n : 0;

    Lvl& lvT = c.lvT; lvT = Lvl(); lvT.nb = B; lvT.max_len = maxT; lvT.total = Ttot; lvT.ld = Ttot;
    const int ffn2_slices = c.ffn2_slices = M.n_layers ? pick_kslices(M.ffn[0].c2, lvT) : 1;
    BufT& bt = c.bt;
    auto layoutT = [&](Arena& A) {
        A.used = 0;
        // one device block mirroring the pinned staging block [geometry ints | length scales | ids | forced durations]:
        // a single host-to-device copy per run
        bt.meta_i = A.get<int>((size_t)9 * B + 8 + B + 2 * (size_t)Ttot);
        bt.ls = (float*)(bt.meta_i + ((size_t)9 * B + 8));
        bt.ids = (int*)(bt.ls + B);
        bt.forced = bt.ids + Ttot;
        bt.x = A.get<float>((size_t)H * Ttot); bt.qkv = A.get<float>((size_t)3 * H * Ttot);
        bt.att = A.get<float>((size_t)H * Ttot); bt.y = A.get<float>((size_t)H * Ttot * ffn2_slices);
        bt.x1 = A.get<float>((size_t)H * Ttot); bt.ffh = A.get<float>((size_t)FF * Ttot);
        bt.m = A.get<float>((size_t)C * Ttot);
        bt.dh = A.get<float>((size_t)(fdp > H ? fdp : H) * Ttot); bt.dt1 = A.get<float>((size_t)fdp * Ttot);
        bt.dt2 = A.get<float>((size_t)fdp * Ttot); bt.dc = A.get<float>((size_t)fdp * Ttot);
        bt.dhh = A.get<float>((size_t)fdp * Ttot); bt.dp29 = A.get<float>((size_t)29 * Ttot);
        for (auto& r : bt.dr) r = A.get<float>(Ttot);
        bt.dlogw = A.get<float>(Ttot);
        bt.dur = A.get<int>(Ttot + B); bt.cum = A.get<int>(Ttot);
        bt.frames = bt.dur + Ttot;     // contiguous with dur: ONE D2H brings both
        bt.g = A.get<float>((size_t)(M.gin > 0 ? M.gin : 1) * B);
        bt.cond_dp = A.get<float>((size_t)(fdp > H ? fdp : H) * B);
        bt.cond_dec = A.get<float>((size_t)(M.up_init > 0 ? M.up_init : 1) * B);
        bt.cond_wn = A.get<float>((size_t)(2 * wnH * wnL + 4) * B * (M.n_flows > 0 ? M.n_flows : 1));   // one block per coupling
    };
    arenaT_.measuring = true; layoutT(arenaT_);
    if (!ensure(arenaT_, arenaT_.used + 4096)) return fail(STS_EDEVICE, "out of device memory (phoneme-level workspace)");
    arenaT_.measuring = false; layoutT(arenaT_);

    // ---------------- one H2D: geometry + ids (+ forced durations)
    const size_t meta_ints = c.meta_ints = (size_t)9 * B + 8;
    const size_t up_bytes = c.up_bytes = (meta_ints + B + 2 * (size_t)Ttot) * 4 + 1024;
    if (!ensure_pinned(up_bytes + ((size_t)Ttot + B) * 4)) return fail(STS_EDEVICE, "pinned host allocation failed");
    int* pm = c.pm = (int*)pinned_;
    int* p_offT = c.p_offT = pm, *p_lenT = c.p_lenT = pm + B, *p_sid = c.p_sid = pm + 2 * B, *p_one = c.p_one = pm + 5 * B;
    c.p_offF = pm + 3 * B; c.p_lenF = pm + 4 * B;
    for (int b = 0; b < B; b++) { p_offT[b] = offT[b]; p_lenT[b] = lenT[b]; p_sid[b] = sid ? sid[b] : 0; }
    p_one[0] = 0; p_one[1] = B;
    if (B == 1) { int* pw0 = pm + 5 * B + 2; c.p_offF[0] = 0; c.p_lenF[0] = 0; pw0[0] = 0; pw0[1] = 0; pw0[2] = 0; }   // (a launch-ahead run: the durations kernel fills the two lengths on the device)
    int* d_offT = c.d_offT = bt.meta_i, *d_lenT = c.d_lenT = bt.meta_i + B, *d_one = c.d_one = bt.meta_i + 5 * B;
    c.d_sid = bt.meta_i + 2 * B; c.d_offF = bt.meta_i + 3 * B; c.d_lenF = bt.meta_i + 4 * B; c.d_win = bt.meta_i + 5 * B + 2;
    float* p_ls = (float*)(pm + meta_ints);
    for (int b = 0; b < B; b++) p_ls[b] = ls ? ls[b] : 1.0f;
    int* p_ids = (int*)(p_ls + B);
    for (int b = 0; b < B; b++) memcpy(p_ids + offT[b], ids[b], sizeof(int) * n[b]);
    int* p_forced = p_ids + Ttot;
    if (have_forced) memcpy(p_forced, forced_dur.data(), sizeof(int) * Ttot);
    HIPCK(hipMemcpyAsync(bt.meta_i, pm, (meta_ints + B + (size_t)Ttot * (have_forced ? 2 : 1)) * 4, hipMemcpyHostToDevice, stream));
    if (B > 1) HIPCK(hipEventRecord(ev_setup_, stream));     // (run_durations, batches launched from the memo: the host rewrites part of this block)

    // single-segment views travel by value (kernels.hpp SegView): no segment-table load in the kernels of a one-utterance call
    static const bool no_inline_seg_knob = exp_flag("STS_NO_INLINE_SEG");   // experiment knob
    const bool no_inline_seg = c.no_inline_seg = no_inline_seg_knob;
    const bool inl = c.inl = B == 1 && !no_inline_seg;
    c.ms = M.is_ms == 1;
    lvT.seg = inl ? SegView{nullptr, nullptr, 1, 0, 0, lenT[0]} : SegView{d_offT, d_lenT, 1, 0, 0, 0};
    Lvl& lvB = c.lvB; lvB = Lvl(); lvB.seg = no_inline_seg ? SegView{d_one, d_one + 1, 1, 0, 0, 0} : SegView{nullptr, nullptr, 1, 0, 0, B};
    lvB.nb = 1; lvB.max_len = B; lvB.total = B; lvB.ld = B;

    return STS_OK;
}

// ---- stage 1: TextEncoder (/root/reference/src/models/TextEncoder.cpp:50-74, attention_encoder.cpp:78-94)
int Engine::run_text_encoder(RunCtx& c) {
    RUN_ALIASES(c)
    // ---------------- TextEncoder (/root/reference/src/models/TextEncoder.cpp:50-74, attention_encoder.cpp:78-94)
    stage_begin(0);
    // The producer of a layer's input x -- the embedding for layer 0, the previous layer's second LayerNorm (which also adds
    // the FFN's split-K partials) afterwards -- runs inside the launch of the first conv that consumes x (col_proj_kernel)
    // where the width is instantiated and the grid is small; otherwise as its own launch.
    static const bool no_colp = exp_flag("STS_NO_COL_LAYER") || exp_flag("STS_NO_COL_PROJ");   // experiment knobs
    // Measured (profiles/r02 notes in docs/HISTORY.md 5b): with a handful of column blocks (one 128-phoneme utterance = 8) a
    // three-pass q/k/v projection makes each of the few workgroups pull 3 x 147 KB of weights through one CU and loses to
    // the separate LayerNorm + chip-wide conv (23 vs 18 us); from a few dozen blocks on it wins (batch 8: -12 us per layer).
    // A single-pass conv (the encoder's output projection) wins at every size.
    const long colp_blocks = (long)((Ttot + 15) / 16);
    const bool colp_grid = (long)((maxT + 15) / 16) * B <= 512;
    auto produce_x_and_conv = [&](int l, const DConv& c, float* y) {
        // l = index of the layer whose input is produced (l == n_layers: the encoder's output projection)
        ColProjArgs pj;
        memset(&pj, 0, sizeof(pj));
        pj.x_out = bt.x; pj.x_ld = Ttot; pj.wc = c.wc; pj.bias = c.bias_rows; pj.Cout = c.Cout; pj.npass = (c.Cout + H - 1) / H;
        pj.y = y; pj.y_ld = Ttot; pj.C = H; pj.seg = lvT.seg; pj.B = B; pj.max_len = maxT;
        if (l == 0) { pj.ids = bt.ids; pj.emb = M.emb; pj.vocab = M.vocab; pj.emb_scale = sqrtf((float)M.hidden); }
        else { pj.a = bt.x1; pj.a_ld = Ttot; pj.bp = bt.y; pj.b_ld = Ttot; pj.nb = ffn2_slices; pj.b_stride = (long)H * Ttot;
               pj.gamma = M.ln2[l - 1].g; pj.beta = M.ln2[l - 1].b; }
        const bool ok = !no_colp && conv_mode != 1 && colp_grid && (pj.npass == 1 || colp_blocks >= 24) && c.k == 1 && c.Cin == H && M.emb_size == H && c.wc &&
                        (l == 0 || M.ln2[l - 1].C == H) && col_proj_eligible(pj);
        if (ok) {
            flops_[0] += 2.0 * c.macs_per_out * (double)Ttot;
            bytes_[0] += 4.0 * ((double)H * Ttot * (l == 0 ? 1.0 : 2.0 + ffn2_slices) + (double)c.Cout * Ttot + (double)H * c.Cout);
            col_proj(pj, stream);
            return;
        }
        if (l == 0) embed(bt.ids, M.emb, M.vocab, M.emb_size, sqrtf((float)M.hidden), bt.x, Ttot, (int)Ttot, stream);
        else ln(M.ln2[l - 1], bt.x1, bt.y, nullptr, bt.x, lvT, 0, 0, ffn2_slices, (long)H * Ttot);
        conv(c, bt.x, lvT, y, lvT, ConvOpt());
    };
    for (int l = 0; l < M.n_layers; l++) {
        const DMha& a = M.mha[l];
        produce_x_and_conv(l, a.qkv, bt.qkv);
        AttnArgs at;
        memset(&at, 0, sizeof(at));
        at.q = bt.qkv; at.k = bt.qkv + (size_t)H * Ttot; at.v = bt.qkv + (size_t)2 * H * Ttot; at.o = bt.att; at.ld = Ttot;
        at.relk = a.relk; at.relv = a.relv; at.kc = a.kc; at.px = a.px; at.win = a.win; at.nheads = 2;
        at.seg = lvT.seg; at.B = B; at.max_len = maxT; at.block_min_wgs = attn_block_min_wgs; at.attn_reg = attn_reg;
        attention(at, stream);
        flops_[0] += 2.0 * 2.0 * (double)a.ch * (double)maxT * (double)Ttot;   // ~ QK^T + PV
        bytes_[0] += 4.0 * 4.0 * (double)H * (double)Ttot;                       // q, k, v in; o out
        {   // output projection + residual + LayerNorm: one launch (col_layer.hip) where the width is instantiated
            static const bool no_col = exp_flag("STS_NO_COL_LAYER");   // experiment knob
            ColLayerArgs g;
            memset(&g, 0, sizeof(g));
            g.x = bt.att; g.x_ld = Ttot; g.wc = a.o.wc; g.bias = a.o.bias;
            g.add = bt.x; g.add_ld = Ttot; g.g2 = M.ln1[l].g; g.b2 = M.ln1[l].b; g.y = bt.x1; g.y_ld = Ttot; g.C = H;
            g.seg = lvT.seg; g.B = B; g.max_len = maxT;
            if (!no_col && conv_mode != 1 && a.o.Cin == H && a.o.Cout == H && a.o.Cin_pad == H && M.ln1[l].C == H && col_layer_eligible(g)) {
                flops_[0] += 2.0 * a.o.macs_per_out * (double)Ttot;
                bytes_[0] += 4.0 * ((double)H * Ttot * 3.0 + (double)H * H);
                col_layer(g, stream);
            } else {
                conv(a.o, bt.att, lvT, bt.y, lvT, ConvOpt());
                ln(M.ln1[l], bt.x, bt.y, nullptr, bt.x1, lvT, 0, 0);
            }
        }
        const DFfn& f = M.ffn[l];
        ConvOpt o1; o1.pad_l = f.ksize == 1 ? 0 : (f.ksize - 1) / 2;
        conv(f.c1, bt.x1, lvT, bt.ffh, lvT, o1);
        ConvOpt o2 = o1; o2.in_act = 1; o2.slope = 0.f;   // nn_relu fused into the consumer's staging
        // few output tiles, long K: split K over workgroups into partial outputs that the LayerNorm below adds up
        o2.kslices = ffn2_slices; o2.kslice_stride = (long)H * Ttot;
        conv(f.c2, bt.ffh, lvT, bt.y, lvT, o2);
        // (the layer's second LayerNorm runs with the next consumer of x: the next layer's q/k/v conv or the output proj)
    }
    produce_x_and_conv(M.n_layers, M.proj, bt.m);
    tap("x_enc", bt.x, H, Ttot, Ttot);
    tap("m", bt.m, C, Ttot, Ttot);
    mark(1);
    return STS_OK;
}

// ---- stage 2: speaker conditioning, duration predictor, durations -> frame counts on the host (the one data-dependent sync)
int Engine::run_durations(RunCtx& c) {
    Model& M = model;
    const int B = c.B; const StreamSpec* const ss = c.ss; const long Ttot = c.Ttot;
    [[maybe_unused]] const int H = c.H, fdp = c.fdp;
    Lvl &lvT = c.lvT, &lvB = c.lvB; BufT& bt = c.bt;
    const size_t up_bytes = c.up_bytes; int* const pm = c.pm; int* const p_offF = c.p_offF; int* const p_lenF = c.p_lenF;
    int* const d_sid = c.d_sid; int* const d_offF = c.d_offF;
    const bool inl = c.inl, ms = c.ms; const int maxT = c.maxT; (void)maxT;

    // ---------------- speaker conditioning vectors (all 1x1 convs on g; SynthesizerTrn.cpp:363-372)
    stage_begin(1);
    if (ms) gather_speaker(M.emb_g, M.spk_num, M.gin, d_sid, B, bt.g, stream);

    // ---------------- duration predictor
    const float* r_final = nullptr;
    if (M.dur_type == 0) {   // /root/reference/src/models/StochasticDurationPredictor.cpp:117-149
        ConvOpt op;
        if (ms) { conv(M.sdp_cond, bt.g, lvB, bt.cond_dp, lvB, ConvOpt()); op.ubias = bt.cond_dp; }
        conv(M.sdp_pre, bt.x, lvT, bt.dh, lvT, op);
        const float* dh = dds(M.sdp_dds, bt.dh, bt.dt1, bt.dt2, lvT);
        conv(M.sdp_proj, dh, lvT, bt.dc, lvT, ConvOpt());
        // the latent z starts as zeros (noise scale 0, StochasticDurationPredictor.cpp:129-131): null inputs stand for it
        const float *r0 = nullptr, *r1 = nullptr;
        float *n0 = bt.dr[2], *n1 = bt.dr[3], *s0 = bt.dr[0], *s1 = bt.dr[1];
        if (M.sdp_flows <= 1) { HIPCK(hipMemsetAsync(bt.dr[0], 0, (size_t)Ttot * 4, stream)); r0 = bt.dr[0]; }
        for (int i = M.sdp_flows - 1; i > 0; i--) {   // flow 0 is skipped; z == 0 because noise_scale == 0
            const DConvFlow& cf = M.cf[i];
            // DDSConv(pre(x0) + g): the 1 -> C input conv and the "+ g" ride inside the first DDSConv layer
            const SplineTail tl{&cf.proj, sqrtf((float)cf.filter), r0, r1, n0, n1};
            bool tail_done = false;
            const float* dhh = dds(cf.dds, bt.dhh, bt.dt1, bt.dt2, lvT, &cf.pre, r0, bt.dc, &tl, &tail_done);
            if (!tail_done) {
                conv(cf.proj, dhh, lvT, bt.dp29, lvT, ConvOpt());
                spline_step(bt.dp29, Ttot, sqrtf((float)cf.filter), r0, r1, n0, n1, Ttot, stream);
            }
            r0 = n0; r1 = n1;
            float* t;
            t = n0; n0 = s0; s0 = t;
            t = n1; n1 = s1; s1 = t;
        }
        r_final = r0;
    } else {                 // /root/reference/src/models/FixDurationPredictor.cpp:75-96
        const float* xin = bt.x;
        if (ms) {
            conv(M.fix_cond, bt.g, lvB, bt.cond_dp, lvB, ConvOpt());
            HIPCK(hipMemcpyAsync(bt.dh, bt.x, (size_t)H * Ttot * 4, hipMemcpyDeviceToDevice, stream));
            add_ubias(bt.dh, Ttot, bt.cond_dp, H, lvT.seg, B, maxT, stream);
            xin = bt.dh;
        }
        conv(M.fix_c1, xin, lvT, bt.dt1, lvT, ConvOpt());
        ln(M.fix_n1, bt.dt1, nullptr, nullptr, bt.dt1, lvT, 1, 0);
        conv(M.fix_c2, bt.dt1, lvT, bt.dt2, lvT, ConvOpt());
        ln(M.fix_n2, bt.dt2, nullptr, nullptr, bt.dt2, lvT, 1, 0);
        conv(M.fix_proj, bt.dt2, lvT, bt.dr[0], lvT, ConvOpt());
        r_final = bt.dr[0];
    }
    // ---------------- the one data-dependent sync: frame counts (+ durations for the API).  The durations kernel writes them
    // straight into host-mapped pinned memory and raises a sequence flag; the host polls that word -- no device-to-host copy
    // kernel and no stream-synchronise wake-up between the duration predictor and the flow (STS_NO_MAPPED_SYNC=1: the copy path)
    static const bool no_mapped = exp_flag("STS_NO_MAPPED_SYNC");
    const size_t hm_ints = (size_t)Ttot + B + 1;
    if (!no_mapped && hm_ints > hmap_cap_) {
        (void)hipStreamSynchronize(stream);
        if (hmap_) (void)hipHostFree(hmap_);
        hmap_ = nullptr; hmap_dev_ = nullptr; hmap_cap_ = 0;
        if (hipHostMalloc((void**)&hmap_, (hm_ints * 2 + 64) * 4, hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer((void**)&hmap_dev_, hmap_, 0) == hipSuccess) { hmap_cap_ = hm_ints * 2 + 64; memset(hmap_, 0, hmap_cap_ * 4); }
        else { if (hmap_) (void)hipHostFree(hmap_); hmap_ = nullptr; hmap_dev_ = nullptr; }
        if (!arrive_ && hipMalloc((void**)&arrive_, 64) == hipSuccess) (void)hipMemsetAsync(arrive_, 0, 64, stream);
    }
    const bool mapped = c.mapped = !no_mapped && hmap_ && arrive_;
    if (mapped) { seq_ = seq_ == 0x7fffffff ? 1 : seq_ + 1; }
    // Launch-ahead (one utterance, plain call): the flow and the decoder are enqueued for a PREDICTED frame capacity and read the real
    // count from device memory, where the durations kernel leaves it (clamped to the capacity).  The host never waits between the duration
    // predictor and the flow; it reads the count after the last kernel is enqueued and sizes the PCM download with it.
    // The prediction is a MEMO, not a guess (round 5): the reference's noise scale is hard-coded 0 (SynthesizerTrn.cpp:357), so the
    // frame count is a pure function of (phoneme ids, speaker, length scale); the engine remembers the count of its last 1 024 distinct (engine.hpp kMemoEntries)
    // requests under a 64-bit hash of exactly those inputs.  A request it has seen runs ahead with the exact count -- never more than the
    // 63 frames of bucket padding the waiting path also carries, never a miss; any other request takes the waiting path.  Should the
    // count land in another 64-frame bucket than predicted (a hash collision), the two stages are repeated the waiting way, so the samples a
    // request returns never depend on what the engine served before (dispatch decisions are taken per bucket).
    // Batches (round 5): the memo is per UTTERANCE.  When every member of a packed batch has been served before, the frame
    // geometry of the whole batch is known on the host before the duration predictor has run: the tables the batched kernels read are
    // uploaded from the memo, flow + decoder are enqueued right behind the durations kernel, and the counts that kernel wrote are compared
    // with the memo after the run's one stream synchronisation (a difference = a hash collision: the two stages are repeated the waiting
    // way).  Identical geometry => identical launches => bit-identical PCM; the host -- a pool or multi-device worker -- no longer
    // spins through the text encoder and the duration predictor.
    long pred = 0;
    c.req_keys.assign(B, 0ull);
    for (int b = 0; b < B; b++) {
        unsigned long long h = 1469598103934665603ull;
        auto mix = [&h](unsigned v) { for (int q = 0; q < 4; q++) { h ^= (v >> (8 * q)) & 0xffu; h *= 1099511628211ull; } };
        mix((unsigned)c.n[b]);
        if (launch_ahead != 2) {          // (2: test mode -- the key is the phoneme count alone, i.e. every request of a length collides)
            mix((unsigned)(c.sid ? c.sid[b] : 0)); { const float lsb = c.ls ? c.ls[b] : 1.0f; unsigned u; memcpy(&u, &lsb, 4); mix(u); }
            for (int t = 0; t < c.n[b]; t++) mix((unsigned)c.ids[b][t]);
        }
        c.req_keys[b] = h | 1ull;
    }
    c.predF.clear();
    if (launch_ahead && !ss && !have_forced && !record_taps && mapped) {
        c.predF.resize(B);
        for (int b = 0; b < B; b++) {
            const auto it = seen_tf_.find(c.req_keys[b]);
            if (it == seen_tf_.end()) { c.predF.clear(); break; }
            c.predF[b] = it->second;
        }
    }
    if (B == 1 && !c.predF.empty()) pred = c.predF[0];
    c.ahead = pred > 0;
    c.ahead_b = B > 1 && !c.predF.empty();
    c.hop = M.hop_total;
    const long cap = c.ahead ? (pred + 63) / 64 * 64 : 0;
    durations(r_final, M.dur_type == 0 ? 1 : 0, M.ea_m, M.ea_logs, bt.ls, have_forced ? bt.forced : nullptr, bt.dlogw,
              bt.dur, bt.cum, bt.frames, lvT.seg, B, stream, mapped ? hmap_dev_ : nullptr, Ttot, seq_, arrive_,
              c.ahead ? c.d_lenF : nullptr, c.ahead ? c.d_win + 2 : nullptr, (int)cap);
    c.forced = have_forced;
    have_forced = false;
    mark(2);
    sync_wait_ms_ = 0;
    if (c.ahead) {
        c.Ftot = cap; c.maxF = (int)cap;                // (capacity; the real count replaces it in run_output)
        return frame_geometry(c);
    }
    if (c.ahead_b) {                                    // the batch's geometry from the memo; checked against the kernel's counts in run_output
        // p_offF / p_lenF live in the pinned block run_setup uploaded: that copy (issued ~0.3 ms of host work ago) must have read them before the
        // host rewrites them (nothing enforced it)
        HIPCK(hipEventSynchronize(ev_setup_));
        c.Ftot = 0; c.maxF = 0;
        for (int b = 0; b < B; b++) {
            const int f = (int)c.predF[b];
            c.p_offF[b] = (int)c.Ftot; c.p_lenF[b] = f; c.Ftot += f; if (f > c.maxF) c.maxF = f;
        }
        return frame_geometry(c);
    }
    int rc = wait_frame_counts(c);
    if (rc != STS_OK) return rc;
    return frame_geometry(c);
}

static inline int wait_class(long phonemes) { int c = 0; while ((1L << (c + 1)) <= phonemes && c < 23) c++; return c; }
void Engine::nap_before_wait(int which, long phonemes) {
    if (!polite_wait) return;
    const WaitEst& e = wait_est_[which][wait_class(phonemes)];
    if (e.us <= 0.0 || e.phonemes <= 0) return;
    const double scale = phonemes < e.phonemes ? (double)phonemes / (double)e.phonemes : 1.0;      // (a smaller request of the class: never oversleep it)
    const double nap = e.us * scale * 0.8 - 70.0;            // (less the timer slack of a normal thread)
    if (nap < 40.0) return;
    struct timespec ts; ts.tv_sec = (time_t)(nap * 1e-6); ts.tv_nsec = (long)((nap - (double)ts.tv_sec * 1e6) * 1e3);
    (void)nanosleep(&ts, nullptr);
}
void Engine::note_wait(int which, long phonemes, double waited_us, bool was_ready_at_wake) {
    if (!polite_wait) return;
    WaitEst& e = wait_est_[which][wait_class(phonemes)];
    // ready at the first look after the nap = the estimate is too long (by an unknown amount): shrink it; otherwise the wait was measured
    if (e.us <= 0.0) { e.us = waited_us * 0.5; e.phonemes = phonemes; }
    else if (was_ready_at_wake) e.us *= 0.85;
    else { e.us = e.us * 0.7 + waited_us * 0.3; e.phonemes = phonemes; }
}
// the run's one stream synchronisation
int Engine::final_sync(long phonemes) {
    if (!polite_wait) { HIPCK(hipStreamSynchronize(stream)); return STS_OK; }
    const double t0 = now_us();
    nap_before_wait(1, phonemes);
    const bool ready = hipStreamQuery(stream) == hipSuccess;
    HIPCK(hipStreamSynchronize(stream));
    note_wait(1, phonemes, now_us() - t0, ready);
    return STS_OK;
}

// blocks until the durations kernel's results are on the host: durations_h, per-utterance frame offsets / counts, Ftot, maxF
int Engine::wait_frame_counts(RunCtx& c) {
    const int B = c.B; const long Ttot = c.Ttot; BufT& bt = c.bt;
    int* const p_offF = c.p_offF; int* const p_lenF = c.p_lenF;
    int* p_down = (int*)(pinned_ + c.up_bytes);
    {
        const auto w0 = std::chrono::steady_clock::now();
        bool polled_unready = true;
        if (c.mapped) {
            volatile int* flag = hmap_;                 // word 0 = sequence flag, then dur[Ttot], frames[B]
            bool ok = false;
            polled_unready = false;
            // (sts_pool / sts_multi workers sleep through most of this wait: engine.hpp polite_wait)
            nap_before_wait(0, Ttot);
            for (long spin = 0; ; spin++) {
                if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq_) { ok = true; polled_unready = spin > 0; break; }
                cpu_relax();
                if (spin >= 20000) std::this_thread::yield();
                if ((spin & 0x3ff) == 0x3ff) {      // every ~1k polls: did the stream die?  (a kernel fault would spin forever)
                    const hipError_t q = hipStreamQuery(stream);
                    if (q == hipSuccess) { ok = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq_; break; }
                    if (q != hipErrorNotReady) return fail(STS_EDEVICE, std::string("stream failed before the frame counts arrived: ") + hipGetErrorString(q));
                    if (std::chrono::steady_clock::now() - w0 > std::chrono::milliseconds(50)) {
                        HIPCK(hipStreamSynchronize(stream));
                        ok = __atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq_;
                        break;
                    }
                }
            }
            if (!ok) return fail(STS_EDEVICE, "frame counts did not arrive");
            p_down = hmap_ + 1;
        } else {
            HIPCK(hipMemcpyAsync(p_down, bt.dur, ((size_t)Ttot + B) * 4, hipMemcpyDeviceToHost, stream));
            HIPCK(hipStreamSynchronize(stream));
        }
        const double waited_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
        sync_wait_ms_ += waited_ms;
        if (c.mapped) note_wait(0, Ttot, waited_ms * 1e3, !polled_unready);
    }
    durations_h.assign(p_down, p_down + Ttot);
    tap("logw", bt.dlogw, 1, Ttot, Ttot);
    c.Ftot = 0; c.maxF = 0;
    for (int b = 0; b < B; b++) {
        int f = p_down[Ttot + b];
        p_offF[b] = (int)c.Ftot; p_lenF[b] = f; c.Ftot += f; if (f > c.maxF) c.maxF = f;
    }
    if (!c.forced && (int)c.req_keys.size() == B)      // the memo of run_durations: every utterance's frame count (replaces, never a running maximum)
        for (int b = 0; b < B; b++) {
            const auto it = seen_tf_.find(c.req_keys[b]);
            if (it != seen_tf_.end()) { it->second = p_lenF[b]; continue; }
            if (seen_order_.size() >= kMemoEntries) { seen_tf_.erase(seen_order_.front()); seen_order_.pop_front(); }
            seen_tf_.emplace(c.req_keys[b], (long)p_lenF[b]);
            seen_order_.push_back(c.req_keys[b]);
        }
    return STS_OK;
}

// frame capacity (one utterance: buckets of 64 frames), geometry tables on the device, room for the PCM download
int Engine::frame_geometry(RunCtx& c) {
    Model& M = model;
    const int B = c.B; const StreamSpec* const ss = c.ss;
    int* const pm = c.pm; int* const p_offF = c.p_offF; int* const p_lenF = c.p_lenF;
    const int hop = c.hop = M.hop_total;
    c.Fld = (B == 1 && !ss) ? (c.Ftot + 63) / 64 * 64 : c.Ftot;
    c.maxFld = (B == 1 && !ss) ? (int)c.Fld : c.maxF;
    if ((double)c.Fld * hop > 2.0e9 || (double)c.Fld * hop * 8 > 6.0e10) return fail(STS_EINVAL, "batch produces too many samples for one call");
    if (!c.ahead) {   // frame geometry + (normal call) the decode windows = the utterances, in the same copy
        int* pw = pm + 5 * B + 2;
        for (int b = 0; b < B; b++) { pw[b] = p_offF[b]; pw[B + b] = p_offF[b]; pw[2 * B + b] = p_lenF[b]; }
        // (a single utterance carries its geometry in the kernel arguments: nothing on the device reads these tables)
        if (!c.inl || ss) HIPCK(hipMemcpyAsync(c.d_offF, p_offF, (size_t)(5 * B + 2) * 4, hipMemcpyHostToDevice, stream));
    }
    h_pcm = nullptr; pcm_in_host_ = false;
    if (host_pcm && !ss) {   // room for the PCM download that rides at the end of this run
        const size_t need = (size_t)c.Fld * hop * 2 + 256;
        if (need > pinned_pcm_cap_) {
            if (pinned_pcm_) (void)hipHostFree(pinned_pcm_);
            pinned_pcm_ = nullptr; pinned_pcm_cap_ = 0;
            pinned_pcm_dev_ = nullptr;
            if (hipHostMalloc((void**)&pinned_pcm_, need + need / 2, hipHostMallocMapped) != hipSuccess) return fail(STS_EDEVICE, "pinned host allocation failed");
            pinned_pcm_cap_ = need + need / 2;
            if (hipHostGetDevicePointer((void**)&pinned_pcm_dev_, pinned_pcm_, 0) != hipSuccess) { pinned_pcm_dev_ = nullptr; (void)hipGetLastError(); }
        }
    }
    return STS_OK;
}

// ---- stage 3: frame-level workspace (flow buffers over all frames, decoder buffers over the decode windows)
int Engine::run_frame_workspace(RunCtx& c) {
    Model& M = model;
    const int B = c.B; const StreamSpec* const ss = c.ss; const int hop = c.hop;
    const long Ftot = c.Fld; const int maxF = c.maxFld;        // (capacity: == the counts except for one utterance, see RunCtx::Fld)
    const int C = c.C, wnH = c.wnH; const bool inl = c.inl;
    int* const p_lenF = c.p_lenF; int* const d_offF = c.d_offF; int* const d_lenF = c.d_lenF;
    // ---------------- frame-level workspace.  The flow works on all Ftot frames; the decoder works on
    // "windows" of z: whole utterances normally (Wcap == Ftot), one chunk plus its two halos when streaming.
    const int halo = c.halo = decoder_halo_frames(M);
    const long Wcap = c.Wcap = ss ? std::min<long>(Ftot, (long)ss->chunk_frames + 2 * halo) : Ftot;
    int upS = 1;
    for (int u : M.up_rate) upS *= u;
    c.upS = upS;
    std::vector<size_t> stage_elems(M.n_up);
    size_t regA = 0, regB = 0;
    { int S = 1; for (int i = 0; i < M.n_up; i++) { S *= M.up_rate[i]; stage_elems[i] = (size_t)(3 + 4 * M.n_resk) * M.ups[i].Cout * Wcap * S;   // upsampler output + 3 per chain (+ the pre-split path: planes / x16 of the stage input, one more per chain)
          if (i & 1) { if (stage_elems[i] > regB) regB = stage_elems[i]; } else { if (stage_elems[i] > regA) regA = stage_elems[i]; } } }
    const long Lsb = c.Lsb = Wcap * upS + B;           // MB-iSTFT: frames + 1 per window
    const int sbC = c.sbC = M.conv_post.Cout;
    // the reverse flow as one launch per WaveNet layer (wn_flow.hip): every coupling must carry the fused operands, with one geometry
    bool use_ff = flow_fused && conv_mode == 0 && conv_math == 3 && !M.cp.empty() && wnH > 0;
    for (size_t i = 0; i < M.cp.size() && use_ff; i++)
        use_ff = M.cp[i].ff.ok && M.cp[i].wn.H == wnH && M.cp[i].wn.n == c.wnL && M.cp[i].ff.G == M.cp[0].ff.G && M.cp[i].wn.in[0].k == M.cp[0].wn.in[0].k;
    if (use_ff) {
        // one round of workgroups at most: a workgroup re-pulls its group's ~290 KB of weights for 32 frames, which pays while the launch
        // is latency-bound (668 frames: 0.31 vs 0.45 ms) and loses by 2.6x once it is not (batch 32: 7.3 vs 2.8 ms; profiles/r04_ab_log.md)
        long tiles = 0;
        if (B == 1) tiles = (Ftot + 31) / 32;
        else for (int b = 0; b < B; b++) tiles += (p_lenF[b] + 31) / 32;
        use_ff = tiles * M.cp[0].ff.G <= 256;
    }
    c.use_ff = use_ff; c.ffG = use_ff ? M.cp[0].ff.G : 0;
    BufF& bf = c.bf;
    auto layoutF = [&](Arena& A) {
        A.used = 0;
        for (int q = 0; q < 2; q++) {
            bf.ff_h[q] = A.get<float>(use_ff ? (size_t)wnH * Ftot : 1);
            bf.ff_part[q] = A.get<float>(use_ff ? (size_t)c.ffG * wnH * Ftot : 1);
            bf.ff_macc[q] = A.get<float>(use_ff ? (size_t)c.ffG * (C / 2) * Ftot : 1);
        }
        bf.ff_alt = A.get<float>(use_ff ? (size_t)C * Ftot : 1);
        bf.z = A.get<float>((size_t)C * Ftot); bf.h = A.get<float>((size_t)wnH * Ftot);
        bf.acts = A.get<float>((size_t)wnH * Ftot); bf.out = A.get<float>((size_t)wnH * Ftot);
        bf.fliptmp = A.get<float>((M.n_flows & 1) ? (size_t)C * Ftot : 1);
        bf.x0 = A.get<float>((size_t)M.up_init * Wcap);
        bf.regA = A.get<float>(regA + 1); bf.regB = A.get<float>(regB + 1);
        if (M.dec_type != 0) {
            bf.tailA = A.get<float>((size_t)sbC * Lsb); bf.tailB = A.get<float>((size_t)sbC * Lsb);
            bf.tailC = A.get<float>((size_t)4 * Wcap * upS * 4);
        } else { bf.tailA = bf.tailB = bf.tailC = nullptr; }
        bf.wave = A.get<float>(record_taps ? (size_t)Wcap * hop : 1);
        bf.pcm = A.get<int16_t>((size_t)Wcap * hop);
    };
    arenaF_.measuring = true; layoutF(arenaF_);
    if (!ensure(arenaF_, arenaF_.used + 4096)) return fail(STS_EDEVICE, "out of device memory (frame-level workspace)");
    arenaF_.measuring = false; layoutF(arenaF_);
    // one short utterance with the PCM wanted on the host: the decoder's last kernel stores its int16 samples into the mapped pinned
    // buffer itself (posted writes over the host link, under the kernel's own run time) instead of a download queued behind it
    if (pcm_direct && host_pcm && !ss && B == 1 && !record_taps && pinned_pcm_dev_ && (size_t)Wcap * hop * 2 + 256 <= pinned_pcm_cap_ &&
        (size_t)Wcap * hop * 2 <= ((size_t)4 << 20)) {
        bf.pcm = pinned_pcm_dev_;
        pcm_in_host_ = true;
    }

    Lvl& lv1 = c.lv1; lv1 = Lvl(); lv1.seg = (inl && !c.ahead) ? SegView{nullptr, nullptr, 1, 0, 0, p_lenF[0]} : SegView{d_offF, d_lenF, 1, 0, 0, 0};
    lv1.nb = B; lv1.max_len = maxF; lv1.total = Ftot; lv1.ld = Ftot;
    mark(7);
    return STS_OK;
}

// ---- stage 4: length regulator + reverse flow
int Engine::run_flow(RunCtx& c) {
    RUN_ALIASES(c)
    const long Fcount = c.Ftot;         // frames of the batch (taps); below, Ftot / maxF are the CAPACITY the buffers and launches are sized for
#define Ftot Fld
#define maxF maxFld
    // ---------------- length regulator (SynthesizerTrn.cpp:304-321, 380-383: z_p == m_expand, noise 0)
    stage_begin(2);
    const int half = C / 2;
    if (use_ff) {
        expand_frames(bt.m, Ttot, bt.cum, lvT.seg, lv1.seg, C, bf.z, Ftot, B, maxF, stream);
        tap("z_p", bf.z, C, Ftot, Fcount);
        const int G = ffG;
        const size_t hF = (size_t)half * Ftot;
        // where the newest version of each half of z lives: z itself, or the alternate buffer (a first layer that applies the previous
        // coupling's pending -m to its x0 half must not write over what its neighbours still read)
        const float* cur[2] = {bf.z, bf.z + hF};
        float* const home[2] = {bf.z, bf.z + hF};
        float* const alt[2] = {bf.ff_alt, bf.ff_alt + hF};
        int pend_set = -1, pend_half = -1;
        auto finish_half = [&](int hh, int set) {       // home[hh] = cur[hh] (+ the pending slices of `set`)
            FlowFinishArgs Fa;
            memset(&Fa, 0, sizeof(Fa));
            Fa.seg = lv1.seg; Fa.B = B; Fa.max_len = maxF; Fa.half = half;
            Fa.src = cur[hh]; Fa.src_ld = Ftot; Fa.dst = home[hh]; Fa.dst_ld = Ftot;
            Fa.macc = set >= 0 ? bf.ff_macc[set] : nullptr; Fa.n = set >= 0 ? G : 0; Fa.macc_stride = (long)hF;
            flow_finish(Fa, stream);
            cur[hh] = home[hh];
        };
        int cs = 0;
        for (int i = M.n_flows - 1; i >= 0; i--, cs ^= 1) {
            const DCoupling& cp = M.cp[i];
            const DWn& w = cp.wn;
            const DFlowFused& ff = cp.ff;
            const int x0h = cp.flipped ? 1 : 0, x1h = 1 - x0h;
            if (pend_set >= 0 && pend_half != x0h) { finish_half(pend_half, pend_set); pend_set = -1; }     // (couplings that do not alternate halves)
            if (w.has_cond) conv(w.cond, bt.g, lvB, bt.cond_wn, lvB, ConvOpt());
            (void)conv_args(cp.pre, nullptr, lv1, nullptr, lv1, ConvOpt(), nullptr);          // the stage's FLOP / byte account, as for the per-conv launches
            for (int l = 0; l < w.n; l++) {
                ConvOpt og; og.epi = EPI_GATE;
                (void)conv_args(w.in[l], nullptr, lv1, nullptr, lv1, og, nullptr);
                ConvOpt orr; orr.epi = EPI_RESSKIP;
                (void)conv_args(w.rs[l], nullptr, lv1, nullptr, lv1, orr, nullptr);
                FlowLayerArgs A;
                memset(&A, 0, sizeof(A));
                A.seg = lv1.seg; A.B = B; A.max_len = maxF; A.tot = Ftot;
                A.H = w.H; A.half = half; A.G = G; A.Cg = ff.Cg; A.k = w.in[l].k; A.halo = (w.in[l].k - 1) / 2;
                A.layer = l;
                if (l == 0) {
                    A.x0 = cur[x0h]; A.x0_ld = Ftot;
                    if (pend_set >= 0) {
                        A.pend = bf.ff_macc[pend_set]; A.pend_n = G; A.pend_stride = (long)hF;
                        A.x0_out = cur[x0h] == home[x0h] ? alt[x0h] : home[x0h]; A.x0_out_ld = Ftot;
                    }
                    A.w_pre = cp.pre.wh2; A.b_pre = cp.pre.bias; A.s_pre = cp.pre.h2_scale;
                    A.w_gate = ff.gate0; A.s_gate = ff.gate0_scale;
                } else {
                    A.h_in = bf.ff_h[(l - 1) & 1]; A.part_in = bf.ff_part[(l - 1) & 1]; A.part_n = G;
                    A.w_gate = w.in[l].wh2; A.s_gate = w.in[l].h2_scale;
                }
                A.part_stride = (long)((size_t)w.H * Ftot);
                A.h_out = bf.ff_h[l & 1]; A.part_out = bf.ff_part[l & 1];
                A.macc = bf.ff_macc[cs]; A.macc_stride = (long)hF; A.macc_init = l == 0;
                A.b_gate = w.in[l].bias;
                if (w.has_cond) { A.ubias = bt.cond_wn + (size_t)l * 2 * w.H * B; A.ubias_ld = B; }
                A.w_c = ff.wc[l]; A.s_c = ff.sc[l]; A.rows_c = ff.rows_c[l]; A.rows_res = ff.rows_res[l];
                A.b_res = ff.b_res[l]; A.b_m = ff.b_m;
                A.ovf = ovf_;
                flow_layer(A, stream);
                if (l == 0 && pend_set >= 0) { cur[x0h] = A.x0_out; pend_set = -1; }
            }
            ConvOpt os; os.epi = EPI_SUB;
            (void)conv_args(cp.post, nullptr, lv1, nullptr, lv1, os, nullptr);
            pend_set = cs; pend_half = x1h;
        }
        if (pend_set >= 0) finish_half(pend_half, pend_set);
        for (int hh = 0; hh < 2; hh++) if (cur[hh] != home[hh]) finish_half(hh, -1);
    } else
    {
    expand_frames(bt.m, Ttot, bt.cum, lvT.seg, lv1.seg, C, bf.z, Ftot, B, maxF, stream);
    tap("z_p", bf.z, C, Ftot, Fcount);

    // ---------------- reverse flow (ResidualCouplingBlock.cpp:59-70, ResidualCouplingLayer.cpp:47-66, WN.cpp:100-149)
    for (int i = M.n_flows - 1; i >= 0; i--) {
        const DCoupling& cp = M.cp[i];
        const float* x0 = bf.z + (size_t)(cp.flipped ? half : 0) * Ftot;
        float* dst = bf.z + (size_t)(cp.flipped ? 0 : half) * Ftot;
        const DWn& w = cp.wn;
        if (w.has_cond) conv(w.cond, bt.g, lvB, bt.cond_wn, lvB, ConvOpt());
        static const int flow_1x1_tile = exp_int("STS_FLOW_1X1_TILE", -1);   // experiment knob
        { ConvOpt op; op.tile = flow_1x1_tile; conv(cp.pre, x0, lv1, bf.h, lv1, op); }
        for (int l = 0; l < w.n; l++) {
            ConvOpt og; og.epi = EPI_GATE;
            if (w.has_cond) og.ubias = bt.cond_wn + (size_t)l * 2 * w.H * B;
            conv(w.in[l], bf.h, lv1, bf.acts, lv1, og);
            ConvOpt orr; orr.epi = EPI_RESSKIP; orr.epi_flag = l == 0 ? 1 : 0; orr.aux = bf.out; orr.tile = flow_1x1_tile;
            conv(w.rs[l], bf.acts, lv1, bf.h, lv1, orr);
        }
        ConvOpt os; os.epi = EPI_SUB; os.tile = flow_1x1_tile;
        conv(cp.post, bf.out, lv1, dst, lv1, os);
    }
    }
    if (M.n_flows & 1) flip_channels(bf.z, Ftot, C, Ftot, bf.fliptmp, stream);
    tap("z", bf.z, C, Ftot, Fcount);
    mark(3);
    return STS_OK;
#undef Ftot
#undef maxF
}


// ---- stage 5: one decode pass over `nw` windows of z
// ---------------- decoder trunk (Generator_hifigan.cpp:139-175 and the identical loops of MS/Istft/MBB)
// One decode pass over `nw` windows of z.  Window w covers frames [zoff[w], zoff[w] + wlen[w]) of the packed z
// and owns the compact range starting at coff[w] in every decoder buffer.  (Normal call: the windows ARE the
// utterances and zoff == coff == offF.)  d_win = device ints {zoff[nw], coff[nw], wlen[nw]}.
// (zoff0 = frame offset of window 0 inside z: the by-value form of a single window)
int Engine::run_decode(RunCtx& c, int nw, long Wtot, int maxW, int zoff0, int wlen0) {
    RUN_ALIASES(c)
    stage_begin(3);
    // (wlen0: real length of a single window, by value; < 0: the windows' geometry is read from the device tables)
    const bool winl = nw == 1 && !no_inline_seg && wlen0 >= 0;
    auto lvF = [&](int scale, int extra) {
        Lvl l; l.seg = winl ? SegView{nullptr, nullptr, scale, extra, 0, wlen0} : SegView{d_win + nw, d_win + 2 * nw, scale, extra, 0, 0};
        l.nb = nw; l.max_len = maxW * scale + extra;
        l.total = Wtot * scale + (long)nw * extra; l.ld = l.total;
        return l;
    };
    Lvl lz; lz.seg = winl ? SegView{nullptr, nullptr, 1, 0, zoff0, wlen0} : SegView{d_win, d_win + 2 * nw, 1, 0, 0, 0};
    lz.nb = nw; lz.max_len = maxW; lz.total = Fld; lz.ld = Fld;
    const Lvl lw1 = lvF(1, 0);
    {
        ConvOpt op;
        if (M.dec_type == 0 && ms) { conv(M.dec_cond, bt.g, lvB, bt.cond_dec, lvB, ConvOpt()); op.ubias = bt.cond_dec; }
        conv(M.conv_pre, bf.z, lz, bf.x0, lw1, op);
    }
    const float* x = bf.x0;
    int S = 1;
    Lvl lx = lw1;
    // the mean over a stage's ResBlock chains is not formed by a launch of its own: the conv that consumes it (the next upsampler, the
    // output conv) adds the chains' outputs up while it stages its input window (ConvArgs::nsum), -4 launches / ~40 us per step
    struct { const float* p[3] = {nullptr, nullptr, nullptr}; int n = 0; float* dst = nullptr; long count = 0; } mean;
    auto with_mean = [&](ConvOpt& o) { if (mean.n >= 2) { o.sum1 = mean.p[1]; o.sum2 = mean.p[2]; o.nsum = mean.n; o.sum_dst = mean.dst; o.sum_n = mean.count; } };
    static const bool no_sum_fold = exp_flag("STS_NO_SUM_FOLD");   // experiment knob
    mark(5);
    in_mfma_region_ = true;
    for (int i = 0; i < M.n_up; i++) {
        const DConv& up = M.ups[i];
        const int S2 = S * M.up_rate[i];
        const Lvl l2 = lvF(S2, 0);
        const size_t ce = (size_t)up.Cout * l2.total;
        float* reg = (i & 1) ? bf.regB : bf.regA;
        float* bup = reg;
        ConvOpt ou; ou.in_act = 1; ou.slope = 0.1f;
        {   // experiment knob: per-stage kernel variant of the upsamplers, e.g. STS_UP_TILE=6--- (digit = conv mode - 2, '-' = automatic)
            static const char* ut = exp_env("STS_UP_TILE");
            if (ut && (int)strlen(ut) > i && ut[i] >= '0' && ut[i] <= '7') ou.tile = ut[i] - '0';
        }
        with_mean(ou);
        conv(up, x, lx, bup, l2, ou);
        mean.n = 0;
        // The nResK ResBlock chains only meet in the final sum (Generator_hifigan.cpp:159-173;
        // /root/reference/src/modules/ResBlock1.cpp:55-69 per chain).
        const int nk = M.n_resk;
        const float* outs[8];
        const int nd0 = nk > 0 ? (int)M.rb[(size_t)i * nk].c1.size() : 0;
        bool grouped = conv_mode == 0 && nk >= 2 && nk <= kMaxGroup;
        for (int j = 0; j < nk && grouped; j++) grouped = (int)M.rb[(size_t)i * nk + j].c1.size() == nd0;
        if (grouped) {   // probe with layer 0 (geometry is the same for every layer of a chain)
            ConvGroup G; G.n = nk;
            const double f0 = flops_[3], b0 = bytes_[3];
            for (int j = 0; j < nk; j++) G.g[j] = conv_args(M.rb[(size_t)i * nk + j].c1[0], bup, l2, bup, l2, ConvOpt(), nullptr);
            flops_[3] = f0; bytes_[3] = b0;
            grouped = conv_group_eligible(G);
        }
        if (grouped) {
            // Layer d of ALL chains goes out as one grouped launch: 2 * nd launches per stage instead of
            // 2 * nd * nResK, nResK times the workgroups per launch (a batch-1 stage otherwise yields only a
            // few hundred), and chains of different kernel size backfill each other inside the grid.
            const float* cur[kMaxGroup];
            for (int j = 0; j < nk; j++) cur[j] = bup;
            // experiment knob STS_CHAIN_STREAMS=<stage mask>: the chains of the masked stages go out as per-chain launches on
            // the prioritised auxiliary streams (heaviest chain first) instead of one grouped launch per layer
            static const int chain_streams_env = exp_int("STS_CHAIN_STREAMS", 0);
            const int chain_streams = chain_streams_dbg >= 0 ? chain_streams_dbg : chain_streams_env;     // (lab: sts_debug_set STS_DBG_CHAIN_STREAMS)
            const bool per_chain = ((chain_streams >> i) & 1) && nk <= kAux;
            int crank[kMaxGroup];
            for (int j = 0; j < nk; j++) {
                crank[j] = 0;
                for (int q = 0; q < nk; q++) {
                    const int kj = M.rb[(size_t)i * nk + j].c1[0].k, kq = M.rb[(size_t)i * nk + q].c1[0].k;
                    if (kq < kj || (kq == kj && q < j)) crank[j]++;
                }
            }
            if (per_chain) {
                (void)hipEventRecord(ev_fork_, stream);
                for (int k = 0; k < kAux; k++) (void)hipStreamWaitEvent(aux_[k], ev_fork_, 0);
            }
            static const bool no_fuse = exp_flag("STS_NO_FUSE");   // experiment knob
            // the pre-split path (conv_h2p.hip) takes a stage only whole: between its layers the chains' tensors live in the x16 layout
            // Measured (profiles/r06_h2p_thresholds_and_two_products.log, r06_h2p_vs_fused128_ab.log).  128 channels: the pre-split pair beats BOTH the
            // staged grouped pair (one HiFi-GAN utterance, 1 002 tiles of 128 x 128: -1.4 % of the trunk) and the fused 128-channel layer kernel
            // (2 ... 64 utterances: -4.6 ... -11 % of the trunk although the fused kernel keeps the intermediate on chip -- its whole-window staging and
            // one-workgroup-per-CU residency cost more than the 8 bytes per value it saves); below ~3 tiles per CU (one MB-iSTFT utterance: 294) the
            // entry split and the second conv's two output tensors cost more than the faster K loop returns.  256+ channels: from ~2 tiles per CU on
            // (one HiFi-GAN utterance = 252 tiles needs the K split over two wave groups the staged kernel has).  h2p (lab): 2 = always, 3 / 4 = always for
            // the 128-channel / the wider stages only.  (The 64-channel stage was tried on this path too: it LOSES 1 % to the fused layer kernel at 32 utterances
            // and 4 % at one -- 4 chunks x k steps are too few to pay for an unfused pair's two epilogues; profiles/r06_h2p_c64_ab.log.)
            const long h2p_tiles = (long)((l2.max_len + 127) / 128) * (up.Cout / 128) * l2.nb * nk;
            const long h2p_min = up.Cout == 128 ? 768 : 512;
            bool h2p_stage = conv_math == 3 && h2p && !per_chain && up.Cout % 128 == 0 && (double)l2.ld * 32.0 < 2.0e9 &&
                             (h2p == 2 || (h2p == 3 && up.Cout == 128) || (h2p == 4 && up.Cout != 128) || h2p_tiles >= h2p_min);
            for (int j = 0; j < nk && h2p_stage; j++) {
                const DResBlock& rb = M.rb[(size_t)i * nk + j];
                for (int d = 0; d < nd0 && h2p_stage; d++) {
                    const DConv &c1 = rb.c1[d], &c2 = rb.c2[d];
                    h2p_stage = c1.wh2p && c2.wh2p && c1.Cin == up.Cout && c1.Cout == up.Cout && c2.Cin == up.Cout && c2.Cout == up.Cout &&
                                (c1.k & 1) && (c2.k & 1) && c1.dil >= 1 && c2.dil >= 1 && c1.pad == c1.dil * (c1.k - 1) / 2 && c2.pad == c2.dil * (c2.k - 1) / 2 &&
                                c1.dil * (c1.k - 1) <= MAX_HALO_H2P && c2.dil * (c2.k - 1) <= MAX_HALO_H2P && !c1.depthwise && !c2.depthwise && !c1.transposed && !c2.transposed;
                }
            }
            for (int d = 0; d < nd0; d++) {
                // narrow stages: the whole layer (conv1 -> lrelu -> conv2 -> + x) of all chains in one launch
                ResLayerGroup R;
                memset(&R, 0, sizeof(R));
                R.n = nk; R.C = up.Cout; R.ld = l2.ld; R.slope = 0.1f; R.seg = l2.seg; R.B = l2.nb; R.max_n = l2.max_len;
                static const int fuse_maxc = exp_int("STS_FUSE_MAXC", 128);   // experiment knob
                bool fuse = !no_fuse && R.C <= fuse_maxc;
                static const bool bf3_nofuse = exp_flag("STS_BF3_NOFUSE");   // experiment knob
                const bool bf3_layer = (conv_math != 1) && !bf3_nofuse;
                // split-bf16 arithmetic: the 64/32-channel stages always run fused; the 128-channel stage (whole window = 147 KB of
                // LDS, one 8-wave workgroup per CU) from ~8 tiles per CU on -- the trunk is power-bound at batch (docs/HISTORY.md 5d), so
                // dropping the intermediate's HBM round trip pays (batch 8: -3 %), while a single utterance's 1 089 tiles on 256
                // workgroup slots only tie the unfused pair
                static const int bf3_fuse128_tiles = exp_int("STS_BF3_FUSE128_TILES", 2048);   // experiment knob
                if (conv_math != 1) {
                    if (bf3_nofuse || R.C > 128) fuse = false;
                    else if (R.C > 64) fuse = fuse && (long)((l2.max_len + 117) / 118) * l2.nb * nk >= bf3_fuse128_tiles;
                }
                for (int j = 0; j < nk && fuse; j++) {
                    const DResBlock& rb = M.rb[(size_t)i * nk + j];
                    const DConv &c1 = rb.c1[d], &c2 = rb.c2[d];
                    fuse = c1.Cin == R.C && c1.Cout == R.C && c2.Cin == R.C && c2.Cout == R.C && c1.Cin_pad == R.C &&
                           c1.Cout_pad == R.C && c2.Cin_pad == R.C && c2.Cout_pad == R.C && !c1.depthwise && !c2.depthwise &&
                           !c1.transposed && !c2.transposed && c2.dil == 1 && c1.pad == c1.dil * (c1.k - 1) / 2 &&
                           c2.pad == (c2.k - 1) / 2;
                    float *t1 = reg + (size_t)(1 + 3 * j) * ce, *pa = t1 + ce, *pb = pa + ce;
                    float* nxt = (cur[j] == pa) ? pb : pa;
                    R.g[j].x = cur[j]; R.g[j].y = nxt; R.g[j].w1 = c1.w; R.g[j].b1 = c1.bias; R.g[j].w2 = c2.w; R.g[j].b2 = c2.bias;
                    R.g[j].k1 = c1.k; R.g[j].dil1 = c1.dil; R.g[j].k2 = c2.k; R.g[j].wu1 = c1.wu; R.g[j].wu2 = c2.wu;
                    R.g[j].wb1 = c1.wb3; R.g[j].wb2 = c2.wb3p;
                    if (conv_math == 3 && c1.wh2 && c2.wh2p) { R.g[j].wb1 = c1.wh2; R.g[j].wb2 = c2.wh2p; R.g[j].ws1 = c1.h2_scale; R.g[j].ws2 = c2.h2_scale; R.math = 1; R.ovf = ovf_; }
                }
                // the 128-channel variant runs 8-wave workgroups, two per CU: only worth it when the grid fills the chip twice
                if (fuse && R.C > 64) fuse = (long)((l2.max_len + 117) / 118) * l2.nb * nk >= 512;
                if (fuse && R.C > 64 && h2p_stage) fuse = false;       // (round 6: the pre-split pair instead of the fused 128-channel layer kernel)
                if (fuse && resblock_layer_eligible(R)) {
                    double fl = 0, flw = 0, f = 0;
                    auto wino_ratio = [](int k) { int n3, n2; wino_split(k, &n3, &n2); return (4.0 * n3 + 3.0 * n2) / (2.0 * k); };
                    for (int j = 0; j < nk; j++) {   // book FLOPs / algorithmic bytes exactly as for the two separate convs
                        const DResBlock& rb = M.rb[(size_t)i * nk + j];
                        ConvOpt o1; o1.in_act = 1; o1.slope = 0.1f;
                        (void)conv_args(rb.c1[d], cur[j], l2, R.g[j].y, l2, o1, &f); fl += f; flw += f * wino_ratio(rb.c1[d].k);
                        ConvOpt o2; o2.in_act = 1; o2.slope = 0.1f; o2.res = cur[j]; o2.epi = EPI_RESADD;
                        (void)conv_args(rb.c2[d], cur[j], l2, R.g[j].y, l2, o2, &f); fl += f; flw += f * wino_ratio(rb.c2[d].k);
                        cur[j] = R.g[j].y;
                    }
                    // both convs in the Winograd domain when the model carries the transformed weights (-31 % MFMAs);
                    // the direct-form fused kernel otherwise
                    static const bool no_wino = exp_flag("STS_NO_WINO");   // experiment knob
                    const bool wino = !no_wino && resblock_wino_eligible(R);
                    if (bf3_layer && resblock_bf3_eligible(R)) {
                        static const int bv = exp_int("STS_BF3_LAYER_VARIANT", -1);   // experiment knob
                        if (per_chain) {
                            for (int j = 0; j < nk; j++) { ResLayerGroup R1 = R; R1.n = 1; R1.g[0] = R.g[j]; resblock_bf3(R1, aux_[crank[j] % kAux], bv); }
                        } else resblock_bf3(R, stream, bv);
                        mfma_flops_ += fl; bf16_exec_ += products() * fl; mfma_launches_ += 1;
                        continue;
                    }
                    if (per_chain) {
                        for (int j = 0; j < nk; j++) {
                            ResLayerGroup R1 = R; R1.n = 1; R1.g[0] = R.g[j];
                            if (wino) resblock_wino(R1, aux_[crank[j] % kAux]); else resblock_layer(R1, aux_[crank[j] % kAux]);
                        }
                    } else if (wino) resblock_wino(R, stream);
                    else resblock_layer(R, stream);
                    mfma_flops_ += fl; mfma_exec_ += wino ? flw : fl; mfma_launches_ += 1;
                    continue;
                }
                // ---- round 6: the wide stages on pre-split, channel-minor activations (conv_h2p.hip).  The stage input is split once
                // (planes P0 of lrelu(x) + the x16 copy X0 for the residual); conv1 reads planes and writes planes of lrelu(out); conv2 reads those
                // and the x16 residual and writes the next layer's x16 + planes -- or, in the last layer, the fp32 [C][ld] tensor the next stage reads.
                // Buffers: t1 (planes of conv1's output), pa / pb (x16 ping-pong, last layer: fp32), pn (planes of the layer output), per chain.
                if (h2p_stage) {
                    H2PGroup H1, H2;
                    memset(&H1, 0, sizeof(H1)); memset(&H2, 0, sizeof(H2));
                    H1.n = H2.n = nk; H1.seg = H2.seg = l2.seg; H1.B = H2.B = l2.nb; H1.max_n = H2.max_n = l2.max_len; H1.ovf = H2.ovf = ovf_;
                    float* const P0 = reg + (size_t)(1 + 4 * nk) * ce; float* const X0 = P0 + ce;
                    double fl = 0, f = 0;
                    for (int j = 0; j < nk; j++) {
                        const DResBlock& rb = M.rb[(size_t)i * nk + j];
                        const DConv &c1 = rb.c1[d], &c2 = rb.c2[d];
                        float *t1 = reg + (size_t)(1 + 3 * j) * ce, *pa = t1 + ce, *pb = pa + ce, *pn = reg + (size_t)(1 + 3 * nk + j) * ce;
                        float* nxt = (cur[j] == pa) ? pb : pa;
                        const bool last = d + 1 == nd0;
                        {   // book FLOPs / algorithmic bytes exactly as for the two staged convs
                            ConvOpt o1; o1.in_act = 1; o1.slope = 0.1f;
                            (void)conv_args(c1, cur[j], l2, t1, l2, o1, &f); fl += f;
                            ConvOpt o2; o2.in_act = 1; o2.slope = 0.1f; o2.res = cur[j]; o2.epi = EPI_RESADD;
                            (void)conv_args(c2, t1, l2, nxt, l2, o2, &f); fl += f;
                        }
                        H2PArgs& a1 = H1.g[j];
                        a1.xp = d == 0 ? (const void*)P0 : (const void*)pn; a1.xp_ld = l2.ld; a1.wb = c1.wh2p; a1.wscale = c1.h2_scale; a1.bias = c1.bias;
                        a1.yp = t1; a1.yp_ld = l2.ld; a1.yp_slope = 0.1f;
                        a1.Cin = c1.Cin; a1.Cout = c1.Cout; a1.ntap = c1.k; a1.tap_step = c1.dil; a1.tap_off = -c1.pad;
                        H2PArgs& a2 = H2.g[j];
                        a2.xp = t1; a2.xp_ld = l2.ld; a2.wb = c2.wh2p; a2.wscale = c2.h2_scale; a2.bias = c2.bias;
                        a2.res16 = d == 0 ? X0 : cur[j]; a2.res_ld = l2.ld;
                        if (last) { a2.y = nxt; a2.y_ld = l2.ld; }
                        else { a2.y16 = nxt; a2.y16_ld = l2.ld; a2.yp = pn; a2.yp_ld = l2.ld; a2.yp_slope = 0.1f; }
                        a2.Cin = c2.Cin; a2.Cout = c2.Cout; a2.ntap = c2.k; a2.tap_step = c2.dil; a2.tap_off = -c2.pad;
                        cur[j] = nxt;
                    }
                    if (d == 0) split_planes(bup, l2.ld, up.Cout, l2.total, 0.1f, P0, X0, l2.ld, ovf_, stream);
                    const int ht = h2p_tile < 0 ? -1 : (up.Cout == 128 ? (h2p_tile & 0xff) : ((h2p_tile >> 8) & 0xff));   // lab: low byte = the 128-channel stage, next = wider ones; 0xff = automatic
                    conv_h2p_group(H1, stream, ht == 0xff ? -1 : ht);
                    conv_h2p_group(H2, stream, ht == 0xff ? -1 : ht);
                    mfma_flops_ += fl; bf16_exec_ += products() * fl; mfma_launches_ += 2;
                    continue;
                }
                ConvGroup G1, G2; G1.n = G2.n = nk;
                double fl1 = 0, fl2 = 0, f = 0;
                for (int j = 0; j < nk; j++) {
                    const DResBlock& rb = M.rb[(size_t)i * nk + j];
                    float *t1 = reg + (size_t)(1 + 3 * j) * ce, *pa = t1 + ce, *pb = pa + ce;
                    ConvOpt o1; o1.in_act = 1; o1.slope = 0.1f;
                    G1.g[j] = conv_args(rb.c1[d], cur[j], l2, t1, l2, o1, &f); fl1 += f;
                    ConvOpt o2; o2.in_act = 1; o2.slope = 0.1f; o2.res = cur[j]; o2.epi = EPI_RESADD;
                    float* nxt = (cur[j] == pa) ? pb : pa;
                    G2.g[j] = conv_args(rb.c2[d], t1, l2, nxt, l2, o2, &f); fl2 += f;
                    cur[j] = nxt;
                }
                static const char* gt = exp_env("STS_GROUP_TILE");   // experiment knob: per-stage tile digits, e.g. "4335"
                const int gtile = gt && (int)strlen(gt) > i ? gt[i] - '0' : -1;
                // every layer is checked on its own: later layers have larger dilations, and a halo beyond the staged
                // LDS window (e.g. k = 11 with dilation 7) must take the per-conv path, which falls back to conv_generic
                if (per_chain) {
                    for (int j = 0; j < nk; j++) {
                        hipStream_t cs = aux_[crank[j] % kAux];
                        for (const ConvArgs* ca : {&G1.g[j], &G2.g[j]}) {
                            if (conv_math != 1 && conv_bf3_eligible(*ca)) conv_bf3(*ca, cs, -1);
                            else if (conv_mfma_eligible(*ca)) conv_mfma(*ca, cs, gtile);
                            else conv_generic(*ca, cs);
                        }
                    }
                    mfma_flops_ += fl1 + fl2; if (conv_math != 1) bf16_exec_ += products() * (fl1 + fl2); else mfma_exec_ += fl1 + fl2; mfma_launches_ += 2;
                    continue;
                }
                if ((conv_math != 1) && conv_bf3_group_eligible(G1) && conv_bf3_group_eligible(G2)) {
                    static const char* bgt = exp_env("STS_BF3_GROUP_TILE");   // experiment knob: per-stage tile digits
                    const int bt = bgt && (int)strlen(bgt) > i ? (bgt[i] >= '0' && bgt[i] <= '9' ? bgt[i] - '0' : (bgt[i] >= 'a' && bgt[i] <= 'z' ? bgt[i] - 'a' + 10 : -1)) : -1;
                    conv_bf3_group(G1, stream, bt);
                    conv_bf3_group(G2, stream, bt);
                    mfma_flops_ += fl1 + fl2; bf16_exec_ += products() * (fl1 + fl2); mfma_launches_ += 2;
                    continue;
                }
                if (conv_group_eligible(G1)) conv_mfma_group(G1, stream, gtile);
                else for (int j = 0; j < nk; j++) { if (conv_mfma_eligible(G1.g[j])) conv_mfma(G1.g[j], stream, -1); else conv_generic(G1.g[j], stream); }
                if (conv_group_eligible(G2)) conv_mfma_group(G2, stream, gtile);
                else for (int j = 0; j < nk; j++) { if (conv_mfma_eligible(G2.g[j])) conv_mfma(G2.g[j], stream, -1); else conv_generic(G2.g[j], stream); }
                mfma_flops_ += fl1 + fl2; mfma_exec_ += fl1 + fl2; mfma_launches_ += 2;
            }
            if (per_chain)
                for (int k = 0; k < kAux; k++) { (void)hipEventRecord(ev_join_[k], aux_[k]); (void)hipStreamWaitEvent(stream, ev_join_[k], 0); }
            for (int j = 0; j < nk; j++) outs[j] = cur[j];
        } else {
            // fallback: the chains run concurrently on separate HIP streams
            const bool fork = nk > 1 && nk <= 8;
            if (fork) {
                (void)hipEventRecord(ev_fork_, stream);
                for (int k = 0; k < kAux; k++) (void)hipStreamWaitEvent(aux_[k], ev_fork_, 0);
            }
            int rank[8];                      // rank[j] = number of chains cheaper than chain j
            for (int j = 0; j < nk && j < 8; j++) {
                rank[j] = 0;
                for (int q = 0; q < nk && q < 8; q++) {
                    const int kj = M.rb[(size_t)i * nk + j].c1[0].k, kq = M.rb[(size_t)i * nk + q].c1[0].k;
                    if (kq < kj || (kq == kj && q < j)) rank[j]++;
                }
            }
            for (int j = 0; j < nk; j++) {
                const DResBlock& rb = M.rb[(size_t)i * nk + j];
                float *t1 = reg + (size_t)(1 + 3 * j) * ce, *pa = t1 + ce, *pb = pa + ce;
                if (fork) cur_ = aux_[rank[j] % kAux];
                const float* cur = bup;
                const int nd = (int)rb.c1.size();
                for (int d = 0; d < nd; d++) {
                    ConvOpt o1; o1.in_act = 1; o1.slope = 0.1f;
                    conv(rb.c1[d], cur, l2, t1, l2, o1);
                    ConvOpt o2; o2.in_act = 1; o2.slope = 0.1f; o2.res = cur; o2.epi = EPI_RESADD;
                    float* nxt = (cur == pa) ? pb : pa;
                    conv(rb.c2[d], t1, l2, nxt, l2, o2);
                    cur = nxt;
                }
                outs[j] = cur;
            }
            if (fork) {
                for (int j = 0; j < nk && j < kAux; j++) { (void)hipEventRecord(ev_join_[j], aux_[j]); (void)hipStreamWaitEvent(stream, ev_join_[j], 0); }
                cur_ = stream;
            }
        }
        // xs = ((rb_0 + rb_1) + ...) / nResK (Generator_hifigan.cpp:159-173): formed by the consumer while it stages its input (2 or 3
        // chains), or -- the fallback -- by a launch that writes it over the (now dead) upsampler output
        if (nk == 1) x = outs[0];                       // (x / 1 == x)
        else if (nk <= 3 && !no_sum_fold) { mean.p[0] = outs[0]; mean.p[1] = outs[1]; mean.p[2] = nk > 2 ? outs[2] : nullptr; mean.n = nk; mean.dst = bup; mean.count = (long)ce; x = outs[0]; }
        else { sum_scale(bup, outs, nk, (long)ce, stream); x = bup; }
        S = S2; lx = l2;
    }
    in_mfma_region_ = false;
    mark(6);

    // ---------------- decoder tail
    float* wave = record_taps ? bf.wave : nullptr;
    const long Ntot = Wtot * hop;
    if (M.dec_type == 0) {          // Generator_hifigan.cpp:177-179 + SynthesizerTrn.cpp:389-396
        ConvOpt o; o.in_act = 1; o.slope = 1e-2f; o.epi = EPI_TANH_PCM; o.pcm = bf.pcm; o.aux = wave;
        with_mean(o);
        conv(M.conv_post, x, lx, nullptr, lx, o);
    } else {                        // Generator_MBB.cpp:174-202, Generator_MS.cpp:198-228, Generator_Istft.cpp:180-197
        const Lvl lsb = lvF(S, 1);
        ConvOpt o; o.in_act = 1; o.slope = 1e-2f; o.reflect = 1;
        with_mean(o);
        conv(M.conv_post, x, lx, bf.tailA, lsb, o);
        const int bands = M.dec_type == 2 ? 1 : 4;
        const Lvl ltm = lvF(S * 4, 0);
        if (tail_fused && M.dec_type != 2 && sbC == 72 && istft_tail_fused_ok(bands, M.fir_taps, M.fir_pad)) {
            // spectrum + inverse DFT / overlap-add + synthesis filter + int16 cast in one launch (misc_kernels.hip istft_tail_fused_kernel)
            const Lvl lo = lvF(S * 16, 0);
            istft_tail_fused(bf.tailA, lsb.ld, lsb.seg, M.synth_fir, M.fir_taps, M.fir_pad, (float)M.subbands, M.fir_bias, wave, bf.pcm, lo.seg, nw, ltm.max_len, stream);
            flops_[3] += 2.0 * (double)Ntot * (16.0 * 4 + 4 * 18 * 4 / 4.0);
            mark(4);
            if (wave) tap("wave", wave, 1, Ntot, (wlen0 >= 0 ? (long)wlen0 : Wtot) * hop);
            return STS_OK;
        }
        istft_spectrum(bf.tailA, lsb.ld, sbC, bf.tailB, lsb.total, stream);
        float* tm = bf.tailC;
        istft_ola(bf.tailB, lsb.ld, bands, 18, lsb.seg, tm, ltm.ld, ltm.seg, nw, ltm.max_len, stream);
        if (M.dec_type == 2) {
            quantize_pcm(tm, bf.pcm, Ntot, stream);
            if (wave) HIPCK(hipMemcpyAsync(wave, tm, (size_t)Ntot * 4, hipMemcpyDeviceToDevice, stream));
        } else {
            const Lvl lo = lvF(S * 16, 0);
            synth_fir(tm, ltm.ld, ltm.seg, M.synth_fir, M.fir_taps, M.fir_pad, (float)M.subbands, M.fir_bias /* 0 unless the blob's learned filter carries one (MS); the PQMF bank has none */, wave, bf.pcm, lo.seg, nw,
                      ltm.max_len, stream);
        }
        flops_[3] += 2.0 * (double)Ntot * (16.0 * 4 + 4 * 18 * 4 / 4.0);
    }
    mark(4);
    if (wave) tap("wave", wave, 1, Ntot, (wlen0 >= 0 ? (long)wlen0 : Wtot) * hop);
    return STS_OK;
}

// The two-term fp16 arithmetic (conv_math 3) cannot hold an activation beyond fp16's range; its kernels raise a word in
// host-mapped memory when they stage one, and the whole utterance batch is then repeated in the split-bf16 form -- results never
// depend on the flag being rare.  A streaming call checks the word before each chunk leaves: raised during the first chunk (or
// before it: text encoder, flow) the call starts over, later only that chunk is decoded again -- split-bf16 from there on.
static constexpr int kRetrySplitBf16 = 1;     // run_once: nothing was handed out, repeat in the split-bf16 form
// An engine that had to repeat two calls in a row stays in the split-bf16 form (a model whose activations do not fit fp16 would
// otherwise pay for both forms on every call) until sts_set_conv_math is called again.
int Engine::run(int B, const int32_t* const* ids, const int32_t* n, const int32_t* sid, const float* ls, const StreamSpec* ss) {
    if (conv_math != 3) return run_once(B, ids, n, sid, ls, ss);
    if (h2_disabled) {
        conv_math = 0;
        const int rc0 = run_once(B, ids, n, sid, ls, ss);
        conv_math = 3;
        return rc0;
    }
    HIPCK(hipSetDevice(device));
    if (!ovf_host_) {
        if (hipHostMalloc((void**)&ovf_host_, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer((void**)&ovf_, ovf_host_, 0) != hipSuccess) {
            if (ovf_host_) (void)hipHostFree(ovf_host_);
            ovf_host_ = nullptr; ovf_ = nullptr;
            return fail(STS_EDEVICE, "host-mapped allocation failed");
        }
    }
    *(volatile unsigned*)ovf_host_ = 0u;
    const bool forced = have_forced;          // (a run consumes the forced durations: the repeat needs them again)
    const long before = h2_fallbacks;
    int rc = run_once(B, ids, n, sid, ls, ss);
    if ((rc == STS_OK && !ss && *(volatile unsigned*)ovf_host_ != 0u) || rc == kRetrySplitBf16) {
        if (h2_fallbacks == 0)
            tts_log(TTS_LOG_WARNING, "summertts_hip: an activation left the fp16 range (|x| > 60000): this call is repeated in the split-bf16 form "
                                   "(same result, about twice the latency; sts_profile.conv_math_fallbacks counts these; STS_CONV_MATH=bf16x3 avoids them)\n");
        h2_fallbacks++;
        conv_math = 0;
        have_forced = forced;
        rc = run_once(B, ids, n, sid, ls, ss);
    }
    conv_math = 3;
    if (h2_fallbacks != before) {
        if (++h2_consecutive >= 2 && !h2_disabled) {
            h2_disabled = true;
            tts_log(TTS_LOG_WARNING, "summertts_hip: two calls in a row had to be repeated: this engine now stays in the split-bf16 form "
                                   "until sts_set_conv_math is called again (sts_profile.conv_math_pinned)\n");
        }
    } else h2_consecutive = 0;
    return rc;
}

int Engine::run_once(int B, const int32_t* const* ids, const int32_t* n, const int32_t* sid, const float* ls, const StreamSpec* ss) {
    Model& M = model;
    if (B <= 0 || !ids || !n) return fail(STS_EINVAL, "empty batch");
    if (ss && (B != 1 || ss->chunk_frames <= 0 || !ss->cb)) return fail(STS_EINVAL, "streaming takes one utterance, a positive chunk size and a callback");
    HIPCK(hipSetDevice(device));
    taps.clear();
    memset(&prof, 0, sizeof(prof));
    for (double& f : flops_) f = 0;
    for (double& f : bytes_) f = 0;
    for (double& f : bytes_w_) f = 0;
    mfma_flops_ = 0; mfma_exec_ = 0; bf16_exec_ = 0; mfma_launches_ = 0; in_mfma_region_ = false;

    host_t0_ = now_us(); host_t_sync_ = 0;
    RunCtx c;
    c.B = B; c.ids = ids; c.n = n; c.sid = sid; c.ls = ls; c.ss = ss;
    int rc;
    if ((rc = run_setup(c)) != STS_OK) return rc;
    mark(0);
    host_us_setup_ = (float)(now_us() - host_t0_); host_us_enq_ = 0;
    if ((rc = run_text_encoder(c)) != STS_OK) return rc;
    if ((rc = run_durations(c)) != STS_OK) return rc;
    if ((rc = run_frame_workspace(c)) != STS_OK) return rc;
    if ((rc = run_flow(c)) != STS_OK) return rc;
    return run_output(c);
}

// ---- stage 6: decode (one pass, or chunk by chunk when streaming), PCM to the host, profile
int Engine::run_output(RunCtx& c) {
    RUN_ALIASES(c)
    n_samples.resize(B);
    if (!ss) {
        // windows = utterances.  One utterance: by value when its frame count is known, from the device tables in a launch-ahead run
        int rc = run_decode(c, B, Fld, maxFld, 0, (B == 1 && !ahead) ? (int)Ftot : -1);
        if (rc != STS_OK) return rc;
        if (ahead) {
            // everything is enqueued.  The PCM download is queued for the CAPACITY (<= 63 frames more than needed) and the run's one
            // stream synchronisation happens before the count is looked at: the host never waits for the count by itself
            if (host_pcm && !pcm_in_host_) HIPCK(hipMemcpyAsync(pinned_pcm_, bf.pcm, (size_t)Fld * hop * 2, hipMemcpyDeviceToHost, stream));
            host_us_enq_ = (float)(now_us() - host_t0_);
            if ((rc = final_sync(c.Ttot)) != STS_OK) return rc;
            host_t_sync_ = now_us();
            if ((rc = wait_frame_counts(c)) != STS_OK) return rc;
            if ((c.Ftot + 63) / 64 * 64 != Fld) {
                // the count is not in the bucket this call was launched for (possible only when two requests share a hash): the output is
                // discarded, and flow + decoder run again the waiting way -- what a request returns does not depend on the engine's history
                ahead_misses++;
                HIPCK(hipStreamSynchronize(stream));
                c.ahead = false;
                flops_[2] = flops_[3] = bytes_[2] = bytes_[3] = bytes_w_[2] = bytes_w_[3] = 0;
                mfma_flops_ = 0; mfma_exec_ = 0; bf16_exec_ = 0; mfma_launches_ = 0; in_mfma_region_ = false;
                if ((rc = frame_geometry(c)) != STS_OK) return rc;
                if ((rc = run_frame_workspace(c)) != STS_OK) return rc;
                if ((rc = run_flow(c)) != STS_OK) return rc;
                return run_output(c);
            }
        }
    }
    const long Fcount = c.Ftot;                // (from here on: the real count)
    for (int b = 0; b < B; b++) n_samples[b] = p_lenF[b] * hop;
    if (!ss) {
        d_pcm = bf.pcm;
        total_samples = Fcount * hop;
        if (host_pcm) {
            if (!ahead && !pcm_in_host_) HIPCK(hipMemcpyAsync(pinned_pcm_, bf.pcm, (size_t)total_samples * 2, hipMemcpyDeviceToHost, stream));
            h_pcm = (const int16_t*)pinned_pcm_;
        }
        if (!ahead) {
            host_us_enq_ = (float)(now_us() - host_t0_);
            { const int rcs = final_sync(c.Ttot); if (rcs != STS_OK) return rcs; }
            host_t_sync_ = now_us();
        }
        if (c.ahead_b) {
            // a batch launched from the memo: the counts the durations kernel wrote (already on the host) must be the remembered ones
            const std::vector<long> pred = c.predF;
            int rc = wait_frame_counts(c);
            if (rc != STS_OK) return rc;
            bool same = true;
            for (int b = 0; b < B; b++) same = same && p_lenF[b] == (int)pred[b];
            c.ahead_b = false;
            if (!same) {
                ahead_misses++;
                flops_[2] = flops_[3] = bytes_[2] = bytes_[3] = bytes_w_[2] = bytes_w_[3] = 0;
                mfma_flops_ = 0; mfma_exec_ = 0; bf16_exec_ = 0; mfma_launches_ = 0; in_mfma_region_ = false;
                if ((rc = frame_geometry(c)) != STS_OK) return rc;
                if ((rc = run_frame_workspace(c)) != STS_OK) return rc;
                if ((rc = run_flow(c)) != STS_OK) return rc;
                return run_output(c);
            }
            prof.launch_ahead = 1;
        }
        HIPCK(hipGetLastError());
    } else {
        // Streaming: chunk c = frames [f0, f1) is decoded from the window [f0 - halo, f1 + halo)
        // clipped to the utterance; the halo covers the decoder's receptive field, so the kept samples are
        // computed from exactly the inputs the one-pass decode sees (bit-identical for a pinned kernel variant).  PCM of a chunk goes to the caller before the next chunk starts.
        if (ms && M.dec_type == 0 && B != 1) return fail(STS_EINVAL, "streaming is single-utterance");
        const long F = Fcount;
        int16_t* hp = nullptr;
        const size_t hp_off = (up_bytes + ((size_t)Ttot + B) * 4 + 255) & ~(size_t)255;
        if (!ensure_pinned(hp_off + (size_t)ss->chunk_frames * hop * 2 + 256)) return fail(STS_EDEVICE, "pinned host allocation failed");
        int* pm = c.pm = (int*)pinned_;
        hp = (int16_t*)(pinned_ + hp_off);
        int* pw = pm + 5 * B + 2;
        d_pcm = nullptr; total_samples = 0;
        for (long f0 = 0; f0 < F; f0 += ss->chunk_frames) {
            const long f1 = std::min<long>(F, f0 + ss->chunk_frames);
            const long w0 = std::max<long>(0, f0 - halo), w1 = std::min<long>(F, f1 + halo);
            pw[0] = (int)w0; pw[1] = 0; pw[2] = (int)(w1 - w0);
            HIPCK(hipMemcpyAsync(d_win, pw, 3 * 4, hipMemcpyHostToDevice, stream));
            const int rc = run_decode(c, 1, w1 - w0, (int)(w1 - w0), (int)w0, (int)(w1 - w0));
            if (rc != STS_OK) return rc;
            const long ns = (f1 - f0) * hop;
            HIPCK(hipMemcpyAsync(hp, bf.pcm + (f0 - w0) * hop, (size_t)ns * 2, hipMemcpyDeviceToHost, stream));
            HIPCK(hipStreamSynchronize(stream));
            if (conv_math == 3 && ovf_host_ && *(volatile unsigned*)ovf_host_ != 0u) {
                if (f0 == 0) return kRetrySplitBf16;          // nothing has left yet (run() repeats the call)
                h2_fallbacks++;
                conv_math = 0;                                 // (run() restores the setting)
                f0 -= ss->chunk_frames;                        // this chunk again
                continue;
            }
            total_samples += ns;
            if (ss->cb(ss->user, hp, (int32_t)ns, (int32_t)(f0 * hop)) != 0) break;
        }
        HIPCK(hipGetLastError());
    }
    const long Ntot = Fcount * hop;
    // one utterance: the launches were sized (and their FLOPs / bytes booked) for the frame capacity; the accounts report the real count
    if (Fld > 0 && Fld != Fcount) {
        const double r = (double)Fcount / (double)Fld;
        flops_[2] *= r; flops_[3] *= r; bytes_[2] *= r; bytes_[3] *= r; mfma_flops_ *= r; mfma_exec_ *= r; bf16_exec_ *= r;
    }
    for (int st = 0; st < 4; st++) bytes_[st] += bytes_w_[st];

    prof.frames = Fcount; prof.samples = Ntot; prof.phonemes = Ttot;
    prof.flops_text_encoder = flops_[0]; prof.flops_duration = flops_[1]; prof.flops_flow = flops_[2]; prof.flops_decoder = flops_[3];
    prof.flops_decoder_mfma = mfma_flops_; prof.decoder_mfma_launches = mfma_launches_; prof.bytes_decoder_min = bytes_[3] + 2.0 * (double)Ntot;
    prof.flops_decoder_mfma_executed = mfma_exec_; prof.flops_decoder_bf16_issued = bf16_exec_;
    prof.conv_math_fallbacks = h2_fallbacks; prof.conv_math_pinned = h2_disabled ? 1 : 0;
    prof.bytes_text_encoder = bytes_[0]; prof.bytes_duration = bytes_[1]; prof.bytes_flow = bytes_[2];
    prof.ms_sync_wait_host = (float)sync_wait_ms_; prof.launch_ahead = (ahead || prof.launch_ahead == 1) ? 1 : 0; prof.launch_ahead_misses = ahead_misses;
    if (profiling == 2) {
        float t = 0;
        (void)hipEventElapsedTime(&t, ev_[5], ev_[6]); prof.ms_decoder_mfma = t;
    } else if (profiling) {
        float t = 0;
        (void)hipEventElapsedTime(&t, ev_[0], ev_[1]); prof.ms_text_encoder = t;
        (void)hipEventElapsedTime(&t, ev_[1], ev_[2]); prof.ms_duration = t;
        (void)hipEventElapsedTime(&t, ev_[7], ev_[3]); prof.ms_flow = t;
        (void)hipEventElapsedTime(&t, ev_[3], ev_[4]); prof.ms_decoder = t;
        (void)hipEventElapsedTime(&t, ev_[5], ev_[6]); prof.ms_decoder_mfma = t;
        prof.ms_total_device = prof.ms_text_encoder + prof.ms_duration + prof.ms_flow + prof.ms_decoder;
    }
    prof.us_host_setup = host_us_setup_; prof.us_host_enqueue = host_us_enq_;
    prof.us_host_tail = host_t_sync_ > 0 ? (float)(now_us() - host_t_sync_) : 0.f;
    return STS_OK;
}

}  // namespace sts
