// kernels.hpp -- device-kernel interface of the MI355X (gfx950) SummerTTS acoustic engine.
//
// Activation layout (same memory order as the reference's column-major Eigen MatrixXf [time, chan],
// /root/reference/src/header/nn_conv1d.h:13-29): channel-major, time contiguous, fp32.  A batch of
// utterances is PACKED along the time axis of one buffer  a[c * ld + pos]; a SegView says where
// utterance b lives so that convolution halos never cross utterance boundaries.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sts {

// Utterance b occupies positions [off[b]*scale + b*extra, ... + len[b]*scale + extra) of a packed row.
// off/len are in "base units" (phonemes for the text-side buffers, frames for the acoustic side);
// scale is the upsampling factor accumulated so far; extra is 1 for the MB-iSTFT tail (frames+1 rows).
// A single-segment view may carry its geometry BY VALUE (off == nullptr: the one segment is {ioff, ilen}): for the
// reference's own call shape -- one utterance per infer() -- every kernel then starts without the dependent scalar load
// of its segment table (one memory round trip less on each of the ~140 latency-bound launches of a batch-1 step).
struct SegView {
    const int* off;
    const int* len;
    int scale;
    int extra;
    int ioff, ilen;
};

enum Epilogue : int {
    EPI_STORE = 0,       // y = v
    EPI_RESADD = 1,      // y = v + res
    EPI_SUB = 3,         // y = y - v                                       (coupling: x1 -= m)
    EPI_GATE = 4,        // y = tanh(v_t) * sigmoid(v_s)                     (WN gated unit)
    EPI_RESSKIP = 5,     // rows <  H: y[row] += v ; rows >= H: aux[row-H] (+)= v   (WN res/skip)
    EPI_TANH_PCM = 6,    // wave = tanh(v) ; pcm = trunc(wave * 32737)       (HiFi-GAN tail, Cout == 1)
};

struct ConvArgs {
    const float* x; long x_ld;      // input  [Cin][x_ld]
    float* y; long y_ld;            // output [Cout][y_ld]
    const float* w;                 // packed weights [ntap][Cin_pad][Cout_pad] (co contiguous); depthwise: [ntap][Cout_pad]
    const float* bias;              // [Cout_pad] (packed row order) or null
    const float* ubias; int ubias_ld;  // per-utterance bias [Cout][ubias_ld = B] or null
    const float* res; long res_ld;  // residual input (EPI_RESADD)
    float* aux; long aux_ld;        // skip output (EPI_RESSKIP) / float wave (EPI_TANH_PCM)
    int16_t* pcm;                   // EPI_TANH_PCM
    int Cin, Cout, Cin_pad, Cout_pad, ntap;
    int tap_step, tap_off;          // input position of output n, tap j: n + j*tap_step + tap_off
    int out_stride, out_off;        // output position of n: n*out_stride + out_off (polyphase transposed conv)
    int transposed, n_extra;        // n runs over [0, transposed ? in_len + n_extra : out_len)
    int depthwise;
    int in_act; float in_slope;     // input activation: 0 none, 1 leaky-relu(slope) (slope 0 == relu)
    int in_reflect;                 // logical input = reflect-pad-left-1 view of x (MB-iSTFT tail)
    int epi, epi_flag; float epi_scale; int H;
    int gate_perm;                  // EPI_GATE rows are packed as (tanh16, sigmoid16) inside every 32-row tile
    SegView in_seg, out_seg;
    int B, max_n;                   // batch size, max n_count over the batch (grid sizing)
    // cross-workgroup split of K (split-K kernel only, EPI_STORE only): slice s of `kslices` accumulates its share of the
    // (tap, channel) groups and writes a PARTIAL output to y + s * kslice_stride (bias rides on slice 0); the consumer
    // (layer_norm: LnArgs::nb) adds the partials up.  For convs with a long K and too few output tiles to fill the chip
    // (FFN second conv at batch 1: K = 2304 over 24 tiles).
    int kslices; long kslice_stride;
    // optional Winograd form of the same weights (conv_wino_*): [seg][4][Cin_pad][Cout_pad], see wino_pack()
    const float* wu; int wino_n3, wino_n2;
    // optional split-bf16 form of the same weights (conv_bf3.hip): every fp32 weight as three bf16 terms, fragment-packed by
    // bf3_pack(); null = this conv only has the fp32 matrix-core path
    const void* wb3;
    // optional restriction of the OUTPUT to the columns n in [keep_lo, keep_hi) (keep_hi == 0: all; plain / residual epilogue of
    // non-polyphase convs only).  No caller sets it since the persistent stage kernel of round 3 was deleted (round 5); the epilogues keep the test
    int keep_lo, keep_hi;
    // arithmetic of the wb3 copy (conv_bf3.hip): 0 = three bf16 terms / six products, 1 = two fp16 terms / three products with the
    // weights scaled by a power of two (wscale = its inverse, applied to the finished tile); ovf (math 1): raised when a staged
    // input value does not fit fp16
    int math; float wscale; unsigned* ovf;
    // optional extra input terms (same layout, ld and segments as x): the logical input is ((x + xs1) [+ xs2]) / nsum for nsum = 2 / 3
    // (0 / 1: plain x) -- the mean over a decoder stage's ResBlock chains (Generator_hifigan.cpp:159-173) formed in the staging code
    // of the conv that consumes it instead of by a launch of its own.  Taken by conv_bf3 (conv_bf3_takes_sum) and conv_cout1 only.
    const float* xs1; const float* xs2; int nsum;
    // polyphase transposed conv with ROW-INTERLEAVED phases (conv_bf3 only; 0 = off, else == out_stride in {2, 4, 8}): the packed rows are
    // rho = cout * stride + phase (Cout / Cout_pad count those merged rows, bias is per merged row), so a lane's four consecutive accumulator
    // rows are consecutive OUTPUT POSITIONS of one channel and the tile leaves through 16-byte (stride 2: 8-byte) stores that fill whole
    // sectors -- the phase-major form writes one dword per lane at a stride of `stride` dwords (round 6, profiles/r06_upsampler_rowph.log)
    int rowph;
};

// Segmented Winograd F(2,3): a k-tap filter is cut into n3 three-tap segments followed by n2 two-tap segments
// (k = 3 n3 + 2 n2, n2 in {0,1,2}); every segment contributes 4 (3-tap) or 3 (2-tap) products per pair of
// outputs instead of 6 / 4, and all segments accumulate into the same four Winograd-domain sums.
__host__ __device__ inline void wino_split(int k, int* n3, int* n2) { *n2 = (k % 3 == 0) ? 0 : (k % 3 == 2 ? 1 : 2); *n3 = (k - 2 * *n2) / 3; }
// w: [Cout][k][Cin] source order (or any accessor order given by the strides), dst: [seg][4][Cin_pad][Cout_pad]
void wino_pack(const float* w, long s_out, long s_tap, long s_in, int Cout, int k, int Cin, int Cin_pad, int Cout_pad, float* dst);
bool conv_wino_eligible(const ConvArgs& a);
void conv_wino(const ConvArgs& a, hipStream_t st);

struct LnArgs {
    const float* a; long a_ld;      // v = a (+ b_0 + b_1 + ... + b_{nb-1}, in this order)
    const float* b; long b_ld;      // b_p = b + p * b_stride (split-K partial outputs of the producing conv); nb <= 1: one operand
    int nb; long b_stride;
    const float* res; long res_ld;  // out = res + f(v) when non-null
    float* y; long y_ld;
    const float* gamma; const float* beta;
    int C;
    int pre_relu, post_gelu;
    // optional fused depthwise conv on the input: v = dw_b[c] + sum_j dw_w[j*dw_ld + c] * a[c][pos + j*dw_dil - dw_pad]
    // (DDSConv's separable conv feeds straight into its LayerNorm, /root/reference/src/modules/DDSConv.cpp:97-100)
    const float* dw_w; const float* dw_b; int dw_k, dw_dil, dw_pad, dw_ld;
    SegView seg; int B, max_len;
};

// Fused column-block layer (col_layer.hip): [depthwise conv + LayerNorm + GELU] -> 1x1 conv (C -> C) -> (+ add) ->
// LayerNorm -> [GELU] -> [+ res], one workgroup per 16 time steps, all C channels.
struct ColLayerArgs {
    const float* x; long x_ld;                  // input [C][ld]
    // optional rank-1 term of the input: h[c][t] = (xs_w[c] * xr[t] + xs_b[c]) + x[c][t] -- the 1 -> C "pre" conv of a ConvFlow
    // (/root/reference/src/modules/ConvFlow.cpp:252-254: pre(x0) + g) folded into the first DDSConv layer; xr == null: xr = 0
    const float* xs_w; const float* xs_b; const float* xr;
    const float* dw_w; const float* dw_b; int dw_k, dw_dil, dw_pad, dw_ld;   // optional depthwise conv (null: plain input)
    const float* g1; const float* b1;           // LayerNorm after the depthwise conv (followed by GELU)
    const float* wc; const float* bias;         // 1x1 conv weights in the kernel's own A-strip order (col_layer_pack)
    const float* add; long add_ld;              // v = add + conv(..)   (null: v = conv(..))
    const float* g2; const float* b2; int post_gelu;
    const float* res; long res_ld;              // out = res + f(v)
    float* y; long y_ld;
    int C;
    SegView seg; int B, max_len;
    // optional tail (the last DDSConv layer of a ConvFlow of the stochastic duration predictor): the layer's output stays in LDS, the
    // 1x1 projection to the 29 spline parameters (/root/reference/src/modules/ConvFlow.cpp:256-258) and the reverse spline step with
    // the channel flip (StochasticDurationPredictor.cpp:136-142: (r0, r1) -> (spline(r1 | h), r0)) run in the same launch; y is not written
    const float* tp_w; const float* tp_b;       // projection weights [C][32] (cin-major, rows >= 29 zero) and bias [29]; null: no tail
    float tp_fs;                                // sqrt(filter channels)
    const float* tp_r0; const float* tp_r1;     // latent halves in (null: zeros)
    float* tp_o0; float* tp_o1;                 // latent halves out
};
bool col_layer_eligible(const ColLayerArgs& a);
bool col_layer_width_ok(int C);
// w: [C out][C in] (blob order of a 1x1 conv) -> dst[C * C] in col_layer_kernel's A-strip order
void col_layer_pack(const float* w, int C, float* dst);
void col_layer(const ColLayerArgs& a, hipStream_t st);

// [embedding | LayerNorm(a + partials)] -> x_out, then y = conv1x1(x_out) with npass * C output rows (col_layer.hip)
struct ColProjArgs {
    const int* ids; const float* emb; int vocab; float emb_scale;        // embedding form (ids != null)
    const float* a; long a_ld; const float* bp; long b_ld; int nb; long b_stride;   // LayerNorm form: v = a + bp_0 + ... + bp_{nb-1}
    const float* gamma; const float* beta;
    float* x_out; long x_ld;                    // the staged input, also written out (residual of the layer's post-LayerNorm)
    const float* wc; const float* bias; int Cout, npass;    // weights in col_proj_pack order, bias in plain row order
    float* y; long y_ld;
    int C;
    SegView seg; int B, max_len;
};
bool col_proj_eligible(const ColProjArgs& a);
void col_proj(const ColProjArgs& a, hipStream_t st);
void col_proj_pack(const float* w, int C, int Cout, float* dst);

struct AttnArgs {
    const float* q; const float* k; const float* v; float* o; long ld;
    const float* relk; const float* relv;   // [kc][px] (reference column-major [px, kc])
    int kc, px, win, nheads;
    SegView seg; int B, max_len;
    int block_min_wgs;              // the 16-queries-per-workgroup matrix-core kernel from this many workgroups on (0: default 96)
    int attn_reg;                   // 1: the one-query kernel with its operands in registers where the shape allows (attention_reg_kernel)
};


// ---- one launch per WaveNet layer of the reverse flow (wn_flow.hip, round 4).  All intermediate tensors are CHANNEL-MINOR [frame][channel]
// (packed frames as everywhere else); group g of G = H / Cg handles the gated channels [g Cg, (g + 1) Cg).
struct FlowLayerArgs {
    SegView seg; int B, max_len; long tot;      // frame-level segments of the packed batch; tot = frames of the whole batch (buffer extents)
    int H, half, G, Cg, k, halo;                // WaveNet width, C / 2, channel groups, gate conv taps and (k - 1) / 2
    int layer;                                  // 0: first layer of a coupling (stages x0, runs `pre` inside the launch)
    // layer 0: x0 rows (channel-major [half][x0_ld]) + pend_n pending -m slices (channel-minor [frame][half], pend_stride floats apart)
    // of the previous coupling; their sum is x0' -- written to x0_out (channel-major, another buffer than x0; null: nothing pending)
    const float* x0; long x0_ld; const float* pend; int pend_n; long pend_stride; float* x0_out; long x0_out_ld;
    const void* w_pre; const float* b_pre; float s_pre;         // `pre` (1x1, half -> H): two-term fp16 pack (natural k order), bias, 2^-s
    // layer > 0: h = h_in + part_in[0 .. part_n)   (channel-minor [frame][H], slices part_stride floats apart)
    const float* h_in; const float* part_in; int part_n; long part_stride;
    float* h_out;                               // this layer's h (each group writes its channel slice of its own columns)
    float* part_out;                            // this layer's partial res sums (group g -> slice g), same strides
    float* macc; long macc_stride; int macc_init;     // -m partial accumulators [frame][half] per group; macc_init: first layer of the coupling
    const void* w_gate; const float* b_gate; float s_gate;      // gate conv, rows (tanh16, sigmoid16) per 32-row tile; layer 0: perm_k order
    const float* ubias; int ubias_ld;           // speaker conditioning of the gate rows [2H][B] or null
    const void* w_c; float s_c; int rows_c, rows_res;           // 1x1 conv on the gated channels (perm_k order, all H channels, chunk-major): rows = [res | -m] padded to 32
    const float* b_res; const float* b_m;       // biases (added once: by group 0; b_m with macc_init)
    unsigned* ovf;                              // raised when a staged value leaves the fp16 range
};
struct FlowFinishArgs {
    SegView seg; int B, max_len; int half;
    const float* src; long src_ld; float* dst; long dst_ld;     // channel-major rows of one half
    const float* macc; int n; long macc_stride;
};
bool flow_layer_shape_ok(int H, int half, int k, int dil, int n_layers);
int flow_layer_groups(int H);
void flow_layer(const FlowLayerArgs& a, hipStream_t st);
void flow_finish(const FlowFinishArgs& a, hipStream_t st);

// ---- launchers (all asynchronous on `st`) ------------------------------------------------------
// Matrix-core (v_mfma_f32_32x32x2_f32) implicit-GEMM conv.  Returns false when the shape is not
// eligible (caller then uses conv_generic).  `tile` < 0 picks a tile configuration heuristically.
bool conv_mfma_eligible(const ConvArgs& a);
void conv_mfma(const ConvArgs& a, hipStream_t st, int tile = -1);
// several independent convs of identical geometry in one grid (decoder ResBlock chains of one stage)
constexpr int kMaxGroup = 4;
struct ConvGroup { ConvArgs g[kMaxGroup]; int n; };
// One ResBlock1 layer, x + conv2(lrelu(conv1_dilated(lrelu(x)))), as ONE kernel for narrow stages
// (C = 32 / 64): the intermediate never leaves the CU (LDS).  Up to kMaxGroup chains per grid.
struct ResLayerArgs {
    const float* x; float* y;                 // [C][ld], distinct buffers
    const float *w1, *b1, *w2, *b2;           // packed [k][C][C] (cout contiguous), biases may be null
    int k1, dil1, k2;
    const float *wu1, *wu2;                   // Winograd-domain copies [seg][4][C][C] (wino_pack) or null
    const void *wb1, *wb2;                    // split-bf16 copies (bf3_pack; wb2 with the k order of the parked intermediate) or null
    float ws1, ws2;                           // ResLayerGroup::math 1: inverse weight scales of wb1 / wb2 (ConvArgs::wscale)
};
struct ResLayerGroup {
    ResLayerArgs g[kMaxGroup];                // must stay the first member (indexed through the kernarg pointer)
    int n, C; long ld; float slope;
    SegView seg; int B; int max_n;
    int math; unsigned* ovf;                  // arithmetic of wb1 / wb2 and its overflow flag (ConvArgs::math / ovf)
};
bool resblock_layer_eligible(const ResLayerGroup& G);
void resblock_layer(const ResLayerGroup& G, hipStream_t st);
// the same layer with both convs in the Winograd domain (segmented F(2,3), see conv_wino_body)
bool resblock_wino_eligible(const ResLayerGroup& G);
void resblock_wino(const ResLayerGroup& G, hipStream_t st);
bool conv_group_eligible(const ConvGroup& G);
// fp32 conv on the bf16 matrix cores: both operands split exactly into three bf16 terms (x = hi + mid + lo, 24 mantissa
// bits), the six products of order <= 2^-16 accumulated in fp32 (conv_bf3.hip)
bool conv_bf3_eligible(const ConvArgs& a);
// an eligible conv whose automatic tile is instantiated with the summed-input staging (ConvArgs::nsum >= 2)
bool conv_bf3_takes_sum(const ConvArgs& a);
bool conv_cout1_takes(const ConvArgs& a);      // conv_generic would route this conv to the single-output-channel FIR kernel
void conv_bf3(const ConvArgs& a, hipStream_t st, int tile = -1);
// workgroups the automatic tile choice would launch (the caller keeps latency-bound launches on the split-K fp32 kernel)
long conv_bf3_blocks(const ConvArgs& a);
bool conv_bf3_group_eligible(const ConvGroup& G);
void conv_bf3_group(const ConvGroup& G, hipStream_t st, int tile = -1);
// wp: packed fp32 weights [nslab][Cin_pad][Cout_pad] (nslab = taps, or phases x taps of a polyphase transposed conv)
// -> dst: [slab-major, see conv_bf3.hip] 3 x bf16; returns the number of bytes written (dst == null: size query)
// perm_k: the 16 input channels of a chunk in the order the fused layer kernel parks its intermediate in (k slot (h, e) of a
// chunk = channel 8 (e >> 2) + 4 h + (e & 3): the accumulator rows one lane holds)
// math 1: the two-term fp16 form of the same layout (three planes P0 / P1 / P2, conv_bf3.hip MATH 1); *wscale receives the inverse
// of the power of two the weights were scaled by
size_t bf3_pack(const float* wp, int nphase, int ntap, int Cin_pad, int Cout_pad, void* dst, bool perm_k = false, int math = 0, float* wscale = nullptr);
// one ResBlock1 layer on the bf16 matrix cores: x + conv2(lrelu(conv1_dilated(lrelu(x)))), whole input window staged once
bool resblock_bf3_eligible(const ResLayerGroup& G);
void resblock_bf3(const ResLayerGroup& G, hipStream_t st, int variant = -1);
void conv_mfma_group(const ConvGroup& G, hipStream_t st, int tile = -1);

// ---- two-term fp16 convs on PRE-SPLIT, channel-minor activations (conv_h2p.hip, round 6): the wide ResBlock stages of the decoder trunk.
// The producer (the split kernel at a stage's entry, then every conv's own epilogue) applies the consumer's leaky relu, splits each value into
// its two fp16 terms (conv_bf3_dev.hpp split8h) and stores them in the order the matrix core reads its operands in, so that a consumer's
// input staging is a plain copy global -> LDS (buffer_load ... lds: no registers, no VALU work) instead of 8 dword loads + ~60 VALU + 2 LDS
// stores per 32 positions x 16 channels.
//   planes  [2 planes][C / 16 chunks][ld positions][2 units][8 fp16]    byte (p, c, t, h) = p * 2 C ld + (c ld + t) 32 + 16 h
//   x16     [C / 16 chunks][ld positions][2 units][8 fp32]               the fp32 value itself (residual operand), same order
//   element e of unit h of chunk c = channel 16 c + 8 (e >> 2) + 4 h + (e & 3)   (the accumulator rows one lane holds: bf3_pack perm_k)
// A tensor in either layout takes 4 C ld bytes -- the size of the fp32 [C][ld] tensor it replaces.
struct H2PArgs {
    const void* xp; long xp_ld;           // input planes (C = Cin)
    const void* wb; float wscale;         // bf3_pack(perm_k = true, math = 1) copy of the weights and the inverse of their power-of-two scale
    const float* bias;                    // [Cout] or null
    const float* res16; long res_ld;      // residual in x16 layout or null
    float* y; long y_ld;                  // fp32 [Cout][y_ld] output or null (the last layer of a chain: what the next stage reads)
    float* y16; long y16_ld;              // x16 output or null (the next layer's residual)
    void* yp; long yp_ld; float yp_slope; // planes of lrelu(out, yp_slope) or null (the next conv's input)
    int Cin, Cout, ntap, tap_step, tap_off;
};
struct H2PGroup { H2PArgs g[kMaxGroup]; int n; SegView seg; int B, max_n; unsigned* ovf; };   // g must stay the first member
bool conv_h2p_group_eligible(const H2PGroup& G);
void conv_h2p_group(const H2PGroup& G, hipStream_t st, int tile = -1);
// ---- the same stages in the Winograd domain (conv_h2w.hip, round 6 lab): segmented F(2,3) / F(2,2), two-term fp16 MFMAs on transformed operands,
// activations as x16 only (raw fp32, channel-minor); the input transform + split happen in registers from an LDS copy of the raw window
struct H2WArgs {
    const float* x16; long x_ld;          // input (C channels); lrelu(in_slope) applied by the kernel (1 = none)
    const void* wu; float wscale;         // bf3_pack(perm_k, math 1) of the h2w_transform_weights pseudo-taps and the inverse of their scale
    const float* bias;
    const float* res16; long res_ld;      // residual (x16) or null
    float* y16; long y16_ld;              // x16 output of lrelu(out, out_slope) (1 = raw) or null
    float* y; long y_ld;                  // fp32 [C][y_ld] output or null
    float in_slope, out_slope;
    int C, k, dil;                        // Cin = Cout = C, "same" padding dil (k - 1) / 2
};
struct H2WGroup { H2WArgs g[kMaxGroup]; int n; SegView seg; int B, max_n; unsigned* ovf; };   // g first (kernarg indexing); members share the dilation
bool conv_h2w_group_eligible(const H2WGroup& G);
void conv_h2w_group(const H2WGroup& G, hipStream_t st);
int h2w_transform_weights(const float* wp, int k, int Cin, int Cout, float* dst);        // wp [k][Cin][Cout] -> dst [4 n3 + 3 n2][Cin][Cout]
void to_x16(const float* x, long x_ld, int C, long n, float* x16, long out_ld, hipStream_t st);
// fp32 [C][ld] -> planes of lrelu(x, slope) (+ the x16 copy when x16 != null); n = positions to convert (the packed total)
void split_planes(const float* x, long x_ld, int C, long n, float slope, void* planes, float* x16, long out_ld, unsigned* ovf, hipStream_t st);
void conv_generic(const ConvArgs& a, hipStream_t st);

void embed(const int* ids, const float* emb, int vocab, int H, float scale, float* x, long ld, int total, hipStream_t st);
void layer_norm(const LnArgs& a, hipStream_t st);
void attention(const AttnArgs& a, hipStream_t st);
// y[c][seg b] += u[c*B + b]
void add_ubias(float* y, long ld, const float* u, int C, SegView seg, int B, int max_len, hipStream_t st);
void gather_speaker(const float* emb_g, int spk_num, int gin, const int* sid, int B, float* g, hipStream_t st);
// y = (((r0 + r1) + r2) + ...) / count  (ResBlock sum, /root/reference/src/models/Generator_hifigan.cpp:159-173)
void sum_scale(float* y, const float* const* r, int count, long n, hipStream_t st);
void flip_channels(float* x, long ld, int C, long n, float* tmp, hipStream_t st);

// SDP spline step: (r0, r1) -> (spline^-1(r1 | h), r0) ; h = [29][ld]
void spline_step(const float* h, long ld, float filter_sqrt, const float* r0, const float* r1,
                 float* o0, float* o1, long n, hipStream_t st);
// logw -> durations.  sdp: logw = (r0 - ea_m) * exp(-ea_logs) (ElementwiseAffine inverse) else logw = r0.
// dur[pos] = forced ? forced[pos] : (int)ceil(exp(logw) * ls[b]);  cum[pos] = inclusive prefix inside the
// utterance;  frames[b] = max(sum, 1).
// host_out (optional): host-MAPPED pinned memory (device pointer), laid out [flag][dur: total][frames: B] (the flag word is never used for data); every workgroup
// also writes its results there and the last one to finish publishes `seq` in the flag word -- the host learns the frame
// counts by polling that word instead of a device-to-host copy + stream synchronisation.  arrive: a zeroed device counter.
void durations(const float* r0, int sdp, float ea_m, float ea_logs, const float* ls, const int* forced,
               float* logw_out, int* dur, int* cum, int* frames, SegView seg, int B, hipStream_t st,
               int* host_out = nullptr, long total = 0, int seq = 0, unsigned* arrive = nullptr,
               int* len_out = nullptr, int* win_len_out = nullptr, int cap = 0);     // optional: frames[b] clamped to cap, into two device tables
// z[c][offF[b] + f] = m[c][offT[b] + phoneme(f)]
void expand_frames(const float* m, long m_ld, const int* cum, SegView segT, SegView segF, int C,
                   float* z, long z_ld, int B, int max_frames, hipStream_t st);

// MB-iSTFT tail.  sb: subband conv output [nb*18 or 18][ld] with per-utterance frames = 16*len+1.
// spec: [bands*18][ld] (re, im per bin) ; tm: [bands][ld4] with per-utterance n = 4*(frames-1).
void istft_spectrum(const float* sb, long ld, int rows, float* spec, long total, hipStream_t st);
void istft_ola(const float* spec, long ld, int bands, int band_rows, SegView seg_frames, float* tm, long tm_ld,
               SegView seg_tm, int B, int max_n, hipStream_t st);
// polyphase synthesis FIR over the x4 zero-stuffed band signals: out[i] = sum_tau sum_b g*tm[b][(i+tau-pad)/4]*fir[tau*4+b]
void synth_fir(const float* tm, long tm_ld, SegView seg_tm, const float* fir, int ntap, int pad, float gain, float bias,
               float* wave, int16_t* pcm, SegView seg_out, int B, int max_n, hipStream_t st);
// istft_spectrum + istft_ola + synth_fir as one launch (4 bands + synthesis filter: the MBB / MS decoders); sb as for istft_spectrum
bool istft_tail_fused_ok(int bands, int ntap, int pad);
void istft_tail_fused(const float* sb, long ld, SegView seg_frames, const float* fir, int ntap, int pad, float gain, float bias, float* wave, int16_t* pcm,
                      SegView seg_out, int B, int max_n, hipStream_t st);
// pcm = (int16)(int32)(wave * 32737)   (SynthesizerTrn.cpp:389-396: truncation, wrap-around)
void quantize_pcm(const float* wave, int16_t* pcm, long n, hipStream_t st);

}  // namespace sts
