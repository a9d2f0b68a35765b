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

template <int NSUB>
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
    if (!live) return;
    const float mean = s / (float)C;
    const float var = sq * (float)(1. / (float)C) - mean * mean;
    const float den = (float)sqrt((double)var + 1e-05);
#pragma unroll
    for (int r = 0; r < 4; r++) {
        float o = ((v[r] - mean) / den) * e_g[r] + e_b[r];
        if (a.post_gelu && !(STS_EXP & 1)) o = gelu_ref(o);
        if (a.res) o = e_res[r] + o;
        a.y[(size_t)(row0 + r) * a.y_ld + base + pos] = o;
    }
}

// A operand layout of col_layer_kernel for a 1x1 conv with C inputs and C outputs: dst[((wave * (C/16) + i4) * 64 + lane) * 4 + e]
// = W[cout = 16 wave + (lane & 15)][cin = 4 (4 i4 + e) + (lane >> 4)];  w is the blob's [out][k = 1][in] order.
void col_layer_pack(const float* w, int C, float* dst) {
    const int nsub = C / 16, kq4 = C / 16;
    for (int wave = 0; wave < nsub; wave++)
        for (int i4 = 0; i4 < kq4; i4++)
            for (int lane = 0; lane < 64; lane++)
                for (int e = 0; e < 4; e++) {
                    const int co = 16 * wave + (lane & 15), ci = 4 * (4 * i4 + e) + (lane >> 4);
                    dst[(((size_t)wave * kq4 + i4) * 64 + lane) * 4 + e] = w[(size_t)co * C + ci];
                }
}
bool col_layer_width_ok(int C) { return C == 32 || C == 64 || C == 192 || C == 256; }   // instantiated widths (16 * NSUB)

bool col_layer_eligible(const ColLayerArgs& a) {
    if (!col_layer_width_ok(a.C)) return false;
    if (!a.wc || !a.g2 || !a.b2 || !a.x || !a.y) return false;
    if (a.dw_w && (!a.g1 || !a.b1 || a.dw_k < 1 || a.dw_k > 3)) return false;
    if (a.xs_w && (!a.dw_w || a.res != a.x)) return false;     // the folded input is defined for the DDSConv form only
    if (a.y == a.x) return false;               // other workgroups read x (halo of the depthwise conv) while this one writes y
    return a.max_len > 0 && a.B > 0;
}

void col_layer(const ColLayerArgs& a, hipStream_t st) {
    const dim3 grid((a.max_len + 15) / 16, a.B);
    switch (a.C) {
        case 32: hipLaunchKernelGGL((col_layer_kernel<2>), grid, dim3(128), 0, st, a); break;
        case 64: hipLaunchKernelGGL((col_layer_kernel<4>), grid, dim3(256), 0, st, a); break;
        case 192: hipLaunchKernelGGL((col_layer_kernel<12>), grid, dim3(768), 0, st, a); break;
        default: hipLaunchKernelGGL((col_layer_kernel<16>), grid, dim3(1024), 0, st, a); break;
    }
}

}  // namespace sts
