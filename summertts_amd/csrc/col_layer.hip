// col_layer.hip -- fused "column block" layers for the latency-bound phoneme-level part of the path.
//
// At one utterance the text encoder and the stochastic duration predictor are ~90 dependent launches of a few
// microseconds each; the time is the launch count, not the arithmetic (profiles/r02_*timeline*.txt: ~4.5 us per trivial
// dependent kernel, ~7 us per small conv).  A LayerNorm needs every channel of a time step, which is why the 1x1 convs
// feeding LayerNorms were separate launches.  Here ONE workgroup owns ALL channels of a block of 16 time steps, so a whole
//     DDSConv layer (/root/reference/src/modules/DDSConv.cpp:84-111)
//         y = gelu(LN1(dwconv_{k,dil}(h)));  z = conv1x1(y);  out = h + gelu(LN2(z))
//     or an attention output projection with its post-LayerNorm (/root/reference/src/modules/attention_encoder.cpp:84-88)
//         out = LN(x + conv1x1(att))
// is a single kernel: three launches (two) become one, and the intermediates never leave the CU.
//
// Mapping: C = 16 * NSUB channels, NSUB waves; wave w owns output rows [16 w, 16 w + 16) of the 1x1 conv as ONE
// v_mfma_f32_16x16x4_f32 accumulator (rows x 16 time steps).  The wave's whole A strip (C / 4 registers: W[ci][row]) is
// requested from L2 before anything else and lands while the input stage (depthwise conv + LayerNorm + GELU, VALU) runs;
// the B operand is the staged [C][16] block in LDS (conflict-free: 64 consecutive floats per MFMA).  LayerNorm
// statistics: 4 rows per lane -> two cross-lane adds -> NSUB per-wave partials through LDS.  The arithmetic follows the
// reference formulas exactly as the unfused kernels do (E[x^2] - mean^2, eps 1e-5 added in double, exp-based GELU).
#include "kernels.hpp"
#include "devmath.hpp"

#ifndef STS_EXP
#define STS_EXP 0   // timing experiments only (tools/exp_build.sh); 0 in every shipped build
#endif

namespace sts {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int cl_seg_start(const SegView& s, int b) { return (s.off ? s.off[b] : s.ioff) * s.scale + b * s.extra; }
__device__ __forceinline__ int cl_seg_len(const SegView& s, int b) { return (s.off ? s.len[b] : s.ilen) * s.scale + s.extra; }

template <int NSUB, bool TAIL = false>   // TAIL: the ConvFlow projection + spline step behind the layer (ColLayerArgs::tp_*)
__global__ __launch_bounds__(64 * NSUB) void col_layer_kernel(ColLayerArgs a) {
    constexpr int C = 16 * NSUB, KQ = C / 4, CG = 4 * NSUB;     // k-steps of 4 channels; channel groups of the input stage
    constexpr int KH = KQ > 48 ? KQ / 2 : KQ;                   // B values kept in registers at a time
    __shared__ float ys[C * 16];                 // staged conv input [C][16]
    __shared__ float red[2][NSUB][16];           // per-wave column partials (sum, sum of squares)
    const int b = blockIdx.y;
    const int len = cl_seg_len(a.seg, b);
    const int n0 = blockIdx.x * 16;
    if (n0 >= len) return;
    const size_t base = (size_t)cl_seg_start(a.seg, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, quad = lane >> 4;
    const int pos = n0 + col;
    const bool live = pos < len;
    const int cg = tid >> 4;                     // input stage: thread (col, cg) owns channels cg + i * CG, i = 0..3

    // ---- 1. every global operand is requested up front, in the order it is needed (loads return in order): the input
    // stage's taps first, then the wave's A strip and the epilogue operands, which land under the input stage's VALU work
    float xin[4][3], dww[4][3], dwb[4], g1[4], b1[4];
    if (a.dw_w) {
        float xrq[3] = {0.f, 0.f, 0.f};          // rank-1 input term: the 1 -> C conv of the flow input, evaluated at the three taps
        if (a.xs_w && a.xr) {
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const int q = pos + j * a.dw_dil - a.dw_pad;
                if (live && j < a.dw_k && q >= 0 && q < len) xrq[j] = a.xr[base + q];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = cg + i * CG;
            const float* row = a.x + (size_t)c * a.x_ld + base;
            g1[i] = a.g1[c]; b1[i] = a.b1[c];
            dwb[i] = a.dw_b ? a.dw_b[c] : 0.f;
            const float sw = a.xs_w ? a.xs_w[c] : 0.f, sb = (a.xs_w && a.xs_b) ? a.xs_b[c] : 0.f;
#pragma unroll
            for (int j = 0; j < 3; j++) {        // dw_k <= 3 here (col_layer_eligible); absent taps carry weight 0
                const int q = pos + j * a.dw_dil - a.dw_pad;
                const bool ok = live && j < a.dw_k && q >= 0 && q < len && !(STS_EXP & 8);
                dww[i][j] = j < a.dw_k ? a.dw_w[(size_t)j * a.dw_ld + c] : 0.f;
                float xv = ok ? row[q] : 0.f;
                if (a.xs_w && ok) xv = __fadd_rn(__fadd_rn(__fmul_rn(sw, xrq[j]), sb), xv);   // (w x + b) + g, rounded as the separate conv did
                xin[i][j] = xv;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) xin[i][0] = live ? a.x[(size_t)(cg + i * CG) * a.x_ld + base + pos] : 0.f;
    }
    // A strip of the wave, W[ci = 4 i + quad][row = 16 wave + col], i = 0..KQ-1: packed by model.hip pack_col so that
    // four consecutive k-steps are one 16-byte load and a wave's load is 1 KB contiguous
    f32x4 af4[KQ / 4];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.wc) + (size_t)wave * (KQ / 4) * 64 + lane;
#pragma unroll
        for (int i = 0; i < KQ / 4; i++) af4[i] = (STS_EXP & 2) ? f32x4{1.f, 1.f, 1.f, 1.f} : wp[(size_t)i * 64];
    }
    const int row0 = wave * 16 + quad * 4;
    float e_bias[4], e_g[4], e_b[4], e_add[4], e_res[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        e_bias[r] = a.bias ? a.bias[row0 + r] : 0.f;
        e_g[r] = a.g2[row0 + r]; e_b[r] = a.b2[row0 + r];
        e_add[r] = (a.add && live) ? a.add[(size_t)(row0 + r) * a.add_ld + base + pos] : 0.f;
        e_res[r] = (a.res && live) ? a.res[(size_t)(row0 + r) * a.res_ld + base + pos] : 0.f;
        if (a.xs_w && a.res && live) {           // the residual is the same folded input h
            const float xr0 = a.xr ? a.xr[base + pos] : 0.f;
            e_res[r] = __fadd_rn(__fadd_rn(__fmul_rn(a.xs_w[row0 + r], xr0), a.xs_b ? a.xs_b[row0 + r] : 0.f), e_res[r]);
        }
    }

    // tail operands (the 29-row projection [C][32]: 8 floats per thread; the latent halves of the 16 time steps)
    f32x4 tw0 = {0.f, 0.f, 0.f, 0.f}, tw1 = {0.f, 0.f, 0.f, 0.f};
    float t_in = 0.f, t_keep = 0.f, t_bias = 0.f;
    if constexpr (TAIL) {
        const f32x4* twp = reinterpret_cast<const f32x4*>(a.tp_w) + tid;
        tw0 = twp[0]; tw1 = twp[64 * NSUB];
        if (tid < 32) t_bias = tid < 29 ? a.tp_b[tid] : 0.f;
        if (tid < 16 && live) { t_in = a.tp_r1 ? a.tp_r1[base + pos] : 0.f; t_keep = a.tp_r0 ? a.tp_r0[base + pos] : 0.f; }   // null = the all-zero latent (noise scale 0)
    }

    // ---- 2. input stage
    if (a.dw_w) {
        float v[4];
        float s = 0.f, sq = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float acc = dwb[i];                  // same accumulation order as layer_norm_kernel's fused depthwise conv
#pragma unroll
            for (int j = 0; j < 3; j++) acc += dww[i][j] * xin[i][j];
            v[i] = acc; s += acc; sq += acc * acc;
        }
        s += __shfl_xor(s, 16, 64); sq += __shfl_xor(sq, 16, 64);
        s += __shfl_xor(s, 32, 64); sq += __shfl_xor(sq, 32, 64);
        if (quad == 0) { red[0][wave][col] = s; red[1][wave][col] = sq; }
        __syncthreads();
        s = 0.f; sq = 0.f;
#pragma unroll
        for (int k = 0; k < NSUB; k++) { s += red[0][k][col]; sq += red[1][k][col]; }
        const float mean = s / (float)C;
        const float var = sq * (float)(1. / (float)C) - mean * mean;
        const float den = (float)sqrt((double)var + 1e-05);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float o = ((v[i] - mean) / den) * g1[i] + b1[i];
            if (!(STS_EXP & 1)) o = gelu_ref(o);
            ys[(cg + i * CG) * 16 + col] = live ? o : 0.f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) ys[(cg + i * CG) * 16 + col] = xin[i][0];
    }
    __syncthreads();

    // ---- 3. 1x1 conv: acc[16 rows x 16 time steps] += W^T[rows][4 ci] * ys[4 ci][16].  The B values are read from LDS in
    // one batch ahead of the MFMAs (no LDS latency inside the chain); two accumulators (even / odd k-steps) keep the
    // dependent-issue latency of the chain off the critical path.
    f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int h0 = 0; h0 < KQ; h0 += KH) {
        float bq[KH];
#pragma unroll
        for (int i = 0; i < KH; i++) bq[i] = ys[(4 * (h0 + i) + quad) * 16 + col];
#pragma unroll
        for (int i = 0; i < ((STS_EXP & 4) ? 2 : KH); i += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af4[(h0 + i) / 4][(h0 + i) % 4], bq[i], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af4[(h0 + i + 1) / 4][(h0 + i + 1) % 4], bq[i + 1], acc1, 0, 0, 0);
        }
    }

    // ---- 4. epilogue: v = add + conv + bias ; LayerNorm over all C rows of the time step ; (gelu) ; (+ res)
    // C/D layout of 16x16x4: column = lane & 15, row = 4 * (lane >> 4) + r
    float v[4];
    float s = 0.f, sq = 0.f;
#pragma unroll
    for (int r = 0; r < 4; r++) { v[r] = e_add[r] + ((acc0[r] + acc1[r]) + e_bias[r]); s += v[r]; sq += v[r] * v[r]; }
    s += __shfl_xor(s, 16, 64); sq += __shfl_xor(sq, 16, 64);
    s += __shfl_xor(s, 32, 64); sq += __shfl_xor(sq, 32, 64);
    // (stage 2's reads of red[] all precede the barrier in front of the K loop: red can be rewritten now)
    if (quad == 0) { red[0][wave][col] = s; red[1][wave][col] = sq; }
    __syncthreads();
    s = 0.f; sq = 0.f;
#pragma unroll
    for (int k = 0; k < NSUB; k++) { s += red[0][k][col]; sq += red[1][k][col]; }
    if (!live && !TAIL) return;
    const float mean = s / (float)C;
    const float var = sq * (float)(1. / (float)C) - mean * mean;
    const float den = (float)sqrt((double)var + 1e-05);
    float o4[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float o = ((v[r] - mean) / den) * e_g[r] + e_b[r];
        if (a.post_gelu && !(STS_EXP & 1)) o = gelu_ref(o);
        if (a.res) o = e_res[r] + o;
        o4[r] = o;
        if (!TAIL) a.y[(size_t)(row0 + r) * a.y_ld + base + pos] = o;
    }
    if constexpr (TAIL) {
    // ---- 5. tail: 29 spline parameters per time step from the layer's output, then the reverse spline step.
    // (every wave's reads of ys precede the barrier of stage 4: the block can be overwritten)
    __shared__ float tp[32 * 16], wb[32];
    __shared__ __attribute__((aligned(16))) float wl[C * 32];
#pragma unroll
    for (int r = 0; r < 4; r++) ys[(row0 + r) * 16 + col] = live ? o4[r] : 0.f;
    reinterpret_cast<f32x4*>(wl)[tid] = tw0; reinterpret_cast<f32x4*>(wl)[tid + 64 * NSUB] = tw1;
    if (tid < 32) wb[tid] = t_bias;
    __syncthreads();
    for (int j = tid >> 4; j < 32; j += 4 * NSUB) {                 // thread (parameter j, time step col): fp32 FMA chain over the C channels
        float acc = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; c++) acc += wl[c * 32 + j] * ys[c * 16 + col];
        acc += wb[j];
        tp[j * 16 + col] = j < 29 ? rq_spline_param(j, acc, a.tp_fs) : 0.f;
    }
    __syncthreads();
    if (tid < 16 && live) {
        float t[29];
#pragma unroll
        for (int j = 0; j < 29; j++) t[j] = tp[j * 16 + tid];
        a.tp_o0[base + pos] = rq_spline_inverse_t(t_in, t);
        a.tp_o1[base + pos] = t_keep;
    }
    }
}

// ------------------------------------------------------------------------------------------------
// [embedding | LayerNorm of (a + split-K partials)] -> 1x1 conv to NPASS * C rows (q/k/v projection, encoder output proj).
// The producer of a text-encoder layer's input and the first conv that consumes it in one launch: the input stage forms
// x (written out: it is the residual of the layer's post-LayerNorm) and stages it in LDS; the conv then runs NPASS passes
// of C rows each over the same staged block, the next pass's A strip in flight under the current pass's MFMA chain.
//   /root/reference/src/modules/attention_encoder.cpp:84-93 (x = LN(x1 + FFN(x1)) -> next layer's q/k/v convs),
//   /root/reference/src/models/TextEncoder.cpp:54-66 (embedding * sqrt(H)) and :68-70 (proj)
// ------------------------------------------------------------------------------------------------
template <int NSUB>
__global__ __launch_bounds__(64 * NSUB) void col_proj_kernel(ColProjArgs a) {
    constexpr int C = 16 * NSUB, KQ = C / 4, CG = 4 * NSUB;
    __shared__ float ys[C * 16];
    __shared__ float red[2][NSUB][16];
    const int b = blockIdx.y;
    const int len = cl_seg_len(a.seg, b);
    const int n0 = blockIdx.x * 16;
    if (n0 >= len) return;
    const size_t base = (size_t)cl_seg_start(a.seg, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, quad = lane >> 4;
    const int pos = n0 + col;
    const bool live = pos < len;
    const int cg = tid >> 4;

    // ---- input-stage operands first, then pass 0's A strip
    float v[4], g[4], be[4];
    if (a.ids) {                                 // embedding lookup (TextEncoder.cpp:54-63; emb_(v, e) = ptr[e * vocab + v])
        int id = live ? a.ids[base + pos] : 0;
        if (id < 0 || id >= a.vocab) id = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = live ? a.emb[(size_t)(cg + i * CG) * a.vocab + id] * a.emb_scale : 0.f;
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int c = cg + i * CG;
            g[i] = a.gamma[c]; be[i] = a.beta[c];
            float x0 = live ? a.a[(size_t)c * a.a_ld + base + pos] : 0.f;
            float pb[8];
#pragma unroll
            for (int q = 0; q < 8; q++) pb[q] = (live && q < a.nb) ? a.bp[(size_t)q * a.b_stride + (size_t)c * a.b_ld + base + pos] : 0.f;
            float bs = pb[0];                    // the partials in a fixed order, exactly as layer_norm_kernel adds them
#pragma unroll
            for (int q = 1; q < 8; q++) bs += pb[q];
            v[i] = x0 + bs;
        }
    }
    f32x4 afa[KQ / 4], afb[KQ / 4];
    auto load_strip = [&](int pass, f32x4 (&dst)[KQ / 4]) {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.wc) + ((size_t)pass * NSUB + wave) * (KQ / 4) * 64 + lane;
#pragma unroll
        for (int i = 0; i < KQ / 4; i++) dst[i] = wp[(size_t)i * 64];
    };
    load_strip(0, afa);

    // ---- input stage
    if (!a.ids) {
        float s = 0.f, sq = 0.f;
#pragma unroll
        for (int i = 0; i < 4; i++) { s += v[i]; sq += v[i] * v[i]; }
        s += __shfl_xor(s, 16, 64); sq += __shfl_xor(sq, 16, 64);
        s += __shfl_xor(s, 32, 64); sq += __shfl_xor(sq, 32, 64);
        if (quad == 0) { red[0][wave][col] = s; red[1][wave][col] = sq; }
        __syncthreads();
        s = 0.f; sq = 0.f;
#pragma unroll
        for (int k = 0; k < NSUB; k++) { s += red[0][k][col]; sq += red[1][k][col]; }
        const float mean = s / (float)C;
        const float var = sq * (float)(1. / (float)C) - mean * mean;
        const float den = (float)sqrt((double)var + 1e-05);
#pragma unroll
        for (int i = 0; i < 4; i++) v[i] = ((v[i] - mean) / den) * g[i] + be[i];
    }
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int c = cg + i * CG;
        ys[c * 16 + col] = live ? v[i] : 0.f;
        if (live) a.x_out[(size_t)c * a.x_ld + base + pos] = v[i];
    }
    __syncthreads();

    // ---- conv: the staged block is read from LDS once and reused by every pass
    float bq[KQ];
#pragma unroll
    for (int i = 0; i < KQ; i++) bq[i] = ys[(4 * i + quad) * 16 + col];
    const int rloc = wave * 16 + quad * 4;
    auto run_pass = [&](int pass, const f32x4 (&af)[KQ / 4]) {
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KQ; i += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i / 4][i % 4], bq[i], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(af[(i + 1) / 4][(i + 1) % 4], bq[i + 1], acc1, 0, 0, 0);
        }
        if (live) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = pass * C + rloc + r;
                if (row < a.Cout) a.y[(size_t)row * a.y_ld + base + pos] = (acc0[r] + acc1[r]) + (a.bias ? a.bias[row] : 0.f);
            }
        }
    };
    for (int pass = 0; pass < a.npass; pass += 2) {          // two strips ping-pong: the next pass's weights are in flight
        if (pass + 1 < a.npass) load_strip(pass + 1, afb);
        run_pass(pass, afa);
        if (pass + 1 < a.npass) {
            if (pass + 2 < a.npass) load_strip(pass + 2, afa);
            run_pass(pass + 1, afb);
        }
    }
}

bool col_proj_eligible(const ColProjArgs& a) {
    if (a.C != 32 && a.C != 64 && a.C != 192) return false;           // instantiated input widths
    if (!a.wc || !a.y || !a.x_out || a.npass < 1 || a.npass > 4 || a.Cout > a.npass * a.C || a.Cout <= (a.npass - 1) * a.C) return false;
    if (a.ids) { if (!a.emb || a.vocab <= 0) return false; }
    else if (!a.a || !a.gamma || !a.beta || a.nb < 0 || a.nb > 8 || (a.nb > 0 && !a.bp)) return false;
    return a.max_len > 0 && a.B > 0;
}

void col_proj(const ColProjArgs& a, hipStream_t st) {
    const dim3 grid((a.max_len + 15) / 16, a.B);
    switch (a.C) {
        case 32: hipLaunchKernelGGL((col_proj_kernel<2>), grid, dim3(128), 0, st, a); break;
        case 64: hipLaunchKernelGGL((col_proj_kernel<4>), grid, dim3(256), 0, st, a); break;
        default: hipLaunchKernelGGL((col_proj_kernel<12>), grid, dim3(768), 0, st, a); break;
    }
}

// A operand layout of col_layer_kernel for a 1x1 conv with C inputs and C outputs: dst[((wave * (C/16) + i4) * 64 + lane) * 4 + e]
// = W[cout = 16 wave + (lane & 15)][cin = 4 (4 i4 + e) + (lane >> 4)];  w is the blob's [out][k = 1][in] order.
void col_layer_pack(const float* w, int C, float* dst) { col_proj_pack(w, C, C, dst); }
// Same order for a 1x1 conv with C inputs and Cout outputs, one C x C block per pass of C output rows (rows >= Cout: zeros):
// dst[(((pass * (C/16) + wave) * (C/16) + i4) * 64 + lane) * 4 + e] = W[cout = pass C + 16 wave + (lane & 15)][cin = 4 (4 i4 + e) + (lane >> 4)]
void col_proj_pack(const float* w, int C, int Cout, float* dst) {
    const int nsub = C / 16, kq4 = C / 16, npass = (Cout + C - 1) / C;
    for (int pass = 0; pass < npass; pass++)
        for (int wave = 0; wave < nsub; wave++)
            for (int i4 = 0; i4 < kq4; i4++)
                for (int lane = 0; lane < 64; lane++)
                    for (int e = 0; e < 4; e++) {
                        const int co = pass * C + 16 * wave + (lane & 15), ci = 4 * (4 * i4 + e) + (lane >> 4);
                        dst[((((size_t)pass * nsub + wave) * kq4 + i4) * 64 + lane) * 4 + e] = co < Cout ? w[(size_t)co * C + ci] : 0.f;
                    }
}
bool col_layer_width_ok(int C) { return C == 32 || C == 64 || C == 192 || C == 256; }   // instantiated widths (16 * NSUB)

bool col_layer_eligible(const ColLayerArgs& a) {
    if (!col_layer_width_ok(a.C)) return false;
    if (!a.wc || !a.g2 || !a.b2 || !a.x || !a.y) return false;
    if (a.dw_w && (!a.g1 || !a.b1 || a.dw_k < 1 || a.dw_k > 3)) return false;
    if (a.xs_w && (!a.dw_w || a.res != a.x)) return false;     // the folded input is defined for the DDSConv form only
    if (a.tp_w && (a.C == 256 || !a.tp_b || !a.tp_o0 || !a.tp_o1 || a.tp_o0 == a.tp_r0 || a.tp_o0 == a.tp_r1 || a.tp_o1 == a.tp_r1)) return false;
    if (a.y == a.x) return false;               // other workgroups read x (halo of the depthwise conv) while this one writes y
    return a.max_len > 0 && a.B > 0;
}

void col_layer(const ColLayerArgs& a, hipStream_t st) {
    const dim3 grid((a.max_len + 15) / 16, a.B);
    if (a.tp_w) {
        switch (a.C) {
            case 32: hipLaunchKernelGGL((col_layer_kernel<2, true>), grid, dim3(128), 0, st, a); break;
            case 64: hipLaunchKernelGGL((col_layer_kernel<4, true>), grid, dim3(256), 0, st, a); break;
            default: hipLaunchKernelGGL((col_layer_kernel<12, true>), grid, dim3(768), 0, st, a); break;
        }
        return;
    }
    switch (a.C) {
        case 32: hipLaunchKernelGGL((col_layer_kernel<2>), grid, dim3(128), 0, st, a); break;
        case 64: hipLaunchKernelGGL((col_layer_kernel<4>), grid, dim3(256), 0, st, a); break;
        case 192: hipLaunchKernelGGL((col_layer_kernel<12>), grid, dim3(768), 0, st, a); break;
        default: hipLaunchKernelGGL((col_layer_kernel<16>), grid, dim3(1024), 0, st, a); break;
    }
}

}  // namespace sts
