// misc_kernels.hip -- the non-convolution kernels of the acoustic path (all HBM/latency bound):
// embedding, LayerNorm, windowed relative-position attention, the duration-side spline flow,
// the length regulator, the MB-iSTFT synthesis tail and the int16 quantiser.  Reductions use
// wave64 shuffles; every kernel reads/writes time-contiguous rows so global accesses coalesce.
#include "kernels.hpp"
#include "devmath.hpp"
#include "knobs.hpp"
#include <stdlib.h>

namespace sts {

__device__ __forceinline__ int seg_start(const SegView& s, int b) { return (s.off ? s.off[b] : s.ioff) * s.scale + b * s.extra; }
__device__ __forceinline__ int seg_len(const SegView& s, int b) { return (s.off ? s.len[b] : s.ilen) * s.scale + s.extra; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------
// embedding * sqrt(hidden)   (/root/reference/src/models/TextEncoder.cpp:54-63; emb_(v,e)=ptr[e*vocab+v])
__global__ void embed_kernel(const int* ids, const float* emb, int vocab, int H, float scale, float* x, long ld, int total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)H * total) return;
    int h = (int)(i / total), pos = (int)(i - (long)h * total);
    int id = ids[pos];
    if (id < 0 || id >= vocab) id = 0;
    x[(size_t)h * ld + pos] = emb[(size_t)h * vocab + id] * scale;
}
void embed(const int* ids, const float* emb, int vocab, int H, float scale, float* x, long ld, int total, hipStream_t st) {
    long n = (long)H * total;
    if (n <= 0) return;
    hipLaunchKernelGGL(embed_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ids, emb, vocab, H, scale, x, ld, total);
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over channels per time step (/root/reference/src/nn_op/nn_layer_norm.cpp:65-86):
// var = sum(x^2)/C - mean^2, eps 1e-5 added in double.  Fused: input add, relu before, gelu after,
// residual add after.
// Workgroup = 32 time steps x 32 channel groups (1024 threads): loads coalesce along time (128-B rows),
// each thread touches only C/32 channels (6 at C = 192) so the kernel is ~2 memory round trips deep;
// the channel sum is combined through LDS.
constexpr int LN_G = 32;
__device__ __forceinline__ float ln_input(const LnArgs& a, int c, size_t p, int pos, int len) {
    float v;
    if (a.dw_w) {       // fused depthwise conv (zero padding inside the utterance)
        v = a.dw_b ? a.dw_b[c] : 0.f;
        const float* row = a.a + (size_t)c * a.a_ld + (p - pos);
        int j = 0;
        for (; j + 2 < a.dw_k; j += 3) {        // 3 taps per pass: independent loads (k = 3 in every known model)
            const int q0 = pos + j * a.dw_dil - a.dw_pad, q1 = q0 + a.dw_dil, q2 = q1 + a.dw_dil;
            const float w0 = a.dw_w[(size_t)j * a.dw_ld + c], w1 = a.dw_w[(size_t)(j + 1) * a.dw_ld + c], w2 = a.dw_w[(size_t)(j + 2) * a.dw_ld + c];
            const float x0 = row[q0 >= 0 && q0 < len ? q0 : pos], x1 = row[q1 >= 0 && q1 < len ? q1 : pos], x2 = row[q2 >= 0 && q2 < len ? q2 : pos];
            v += w0 * ((q0 >= 0 && q0 < len) ? x0 : 0.f);
            v += w1 * ((q1 >= 0 && q1 < len) ? x1 : 0.f);
            v += w2 * ((q2 >= 0 && q2 < len) ? x2 : 0.f);
        }
        for (; j < a.dw_k; j++) {
            const int q = pos + j * a.dw_dil - a.dw_pad;
            if (q >= 0 && q < len) v += a.dw_w[(size_t)j * a.dw_ld + c] * row[q];
        }
    } else {
        v = a.a[(size_t)c * a.a_ld + p];
    }
    if (a.b) {
        if (a.nb > 1) {     // split-K partials: all loads issued together, summed in a fixed order (slices beyond nb add +0)
            float pb[8];
#pragma unroll
            for (int q = 0; q < 8; q++) pb[q] = q < a.nb ? a.b[(size_t)q * a.b_stride + (size_t)c * a.b_ld + p] : 0.f;
            float bs = pb[0];
#pragma unroll
            for (int q = 1; q < 8; q++) bs += pb[q];
            v += bs;
        } else {
            v += a.b[(size_t)c * a.b_ld + p];
        }
    }
    if (a.pre_relu && v < 0.f) v = 0.f;
    return v;
}
__global__ __launch_bounds__(1024) void layer_norm_kernel(LnArgs a) {
    __shared__ float rs[LN_G][33], rq[LN_G][33];
    const int b = blockIdx.y;
    const int len = seg_len(a.seg, b);
    const int tx = threadIdx.x & 31, cy = threadIdx.x >> 5;
    const int pos = blockIdx.x * 32 + tx;
    const bool live = pos < len;
    const size_t p = (size_t)seg_start(a.seg, b) + (live ? pos : 0);
    constexpr int MAXV = 8;                 // values kept in registers between the two passes (C <= 256)
    float vals[MAXV], gam[MAXV], bet[MAXV], rsd[MAXV];
    float s = 0.f, sq = 0.f;
    if (live) {
        // affine parameters and the residual do not depend on the statistics: their loads are issued together
        // with the input's, so the kernel has ONE memory round trip before the reduction instead of two
#pragma unroll
        for (int u = 0; u < MAXV; u++) {
            const int c = cy + u * LN_G;
            if (c < a.C) { gam[u] = a.gamma[c]; bet[u] = a.beta[c]; rsd[u] = a.res ? a.res[(size_t)c * a.res_ld + p] : 0.f; }
        }
#pragma unroll
        for (int u = 0; u < MAXV; u++) {
            const int c = cy + u * LN_G;
            if (c < a.C) { const float v = ln_input(a, c, p, pos, len); vals[u] = v; s += v; sq += v * v; }
        }
        for (int c = cy + MAXV * LN_G; c < a.C; c += LN_G) { const float v = ln_input(a, c, p, pos, len); s += v; sq += v * v; }
    }
    rs[cy][tx] = s; rq[cy][tx] = sq;
    __syncthreads();
    if (!live) return;
    s = 0.f; sq = 0.f;
#pragma unroll
    for (int k = 0; k < LN_G; k++) { s += rs[k][tx]; sq += rq[k][tx]; }
    const float mean = s / (float)a.C;
    const float scale = (float)(1. / (float)a.C);
    const float var = sq * scale - mean * mean;
    const float den = (float)sqrt((double)var + 1e-05);
    auto finish = [&](int c, float v) {
        float o = ((v - mean) / den) * a.gamma[c] + a.beta[c];
        if (a.post_gelu) o = gelu_ref(o);
        if (a.res) o = a.res[(size_t)c * a.res_ld + p] + o;
        a.y[(size_t)c * a.y_ld + p] = o;
    };
#pragma unroll
    for (int u = 0; u < MAXV; u++) {
        const int c = cy + u * LN_G;
        if (c < a.C) {
            float o = ((vals[u] - mean) / den) * gam[u] + bet[u];
            if (a.post_gelu) o = gelu_ref(o);
            if (a.res) o = rsd[u] + o;
            a.y[(size_t)c * a.y_ld + p] = o;
        }
    }
    for (int c = cy + MAXV * LN_G; c < a.C; c += LN_G) finish(c, ln_input(a, c, p, pos, len));
}
void layer_norm(const LnArgs& a, hipStream_t st) {
    if (a.max_len <= 0 || a.B <= 0) return;
    hipLaunchKernelGGL(layer_norm_kernel, dim3((a.max_len + 31) / 32, a.B), dim3(1024), 0, st, a);
}

// ---------------------------------------------------------------------------------------------
// Windowed relative-position self-attention, one workgroup per (query i, head, utterance).
// /root/reference/src/modules/multi_head_attention.cpp:201-295 with the skew/pad/reshape of
// :133-199 reduced to its banded meaning:  S[i][j] = (q_i/sqrt(kc)) . k_j + [|j-i|<=win] (q_i/sqrt(kc)) . relK[j-i+win]
// P = exp(S)/sum exp(S) (nn_softmax.cpp:5-28: no max shift) ;  O_i = sum_j P_ij v_j + sum_{|j-i|<=win} P_ij relV[j-i+win].
// Parallel layout: the 4 waves split the kc channels of the score dot product (partials in LDS), lanes run
// over keys; the P.V product runs 4 channels per lane-pass so 4 loads are in flight per wave.
__global__ __launch_bounds__(256) void attention_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.z, h = blockIdx.y, i = blockIdx.x;
    const int T = seg_len(a.seg, b);
    if (i >= T) return;
    const size_t base = (size_t)seg_start(a.seg, b);
    const int kc = a.kc, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* qs = sm;                 // [kc]
    float* red = sm + kc;           // [8]
    float* P = sm + kc + 8;         // [T]
    float* part = P + T;            // [4][T] partial scores
    const float sq = sqrtf((float)kc);
    for (int c = tid; c < kc; c += 256) qs[c] = a.q[(size_t)(h * kc + c) * a.ld + base + i] / sq;
    __syncthreads();
    // banded relative-key logits q . relK[r], r = 0..px-1: one 16-lane group per r, shuffle-reduced
    float* qrel = part + 4 * T;      // [px]
    if (a.win > 0) {
        const int l16 = tid & 15;
        for (int r = tid >> 4; r < a.px; r += 16) {   // 16 groups of 16 lanes; any window size (px = 2 win + 1)
            float s = 0.f;
            for (int c = l16; c < kc; c += 16) s += qs[c] * a.relk[(size_t)c * a.px + r];
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 16);
            if (l16 == 0) qrel[r] = s;
        }
    }
    // wave w owns channels [c0, c1)
    const int cw = (kc + 3) / 4, c0 = wave * cw, c1 = c0 + cw < kc ? c0 + cw : kc;
    for (int j = lane; j < T; j += 64) {
        const float* kp = a.k + (size_t)(h * kc) * a.ld + base + j;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int c = c0;
#pragma unroll 2
        for (; c + 3 < c1; c += 4) {
            s0 += qs[c] * kp[(size_t)c * a.ld];
            s1 += qs[c + 1] * kp[(size_t)(c + 1) * a.ld];
            s2 += qs[c + 2] * kp[(size_t)(c + 2) * a.ld];
            s3 += qs[c + 3] * kp[(size_t)(c + 3) * a.ld];
        }
        for (; c < c1; c++) s0 += qs[c] * kp[(size_t)c * a.ld];
        part[wave * T + j] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    float psum = 0.f;
    for (int j = tid; j < T; j += 256) {
        float sc = (part[j] + part[T + j]) + (part[2 * T + j] + part[3 * T + j]);
        const int r = j - i + a.win;
        if (a.win > 0 && r >= 0 && r < a.px) sc += qrel[r];
        const float e = expf(sc);
        P[j] = e;
        psum += e;
    }
    psum = wave_sum(psum);
    if (lane == 0) red[wave] = psum;
    __syncthreads();
    const float sum = red[0] + red[1] + red[2] + red[3];
    for (int j = tid; j < T; j += 256) P[j] = P[j] / sum;
    __syncthreads();
    // P.V: one thread per (channel, half of the keys): every load is independent (deep pipelining, no
    // cross-lane reduction per channel); the two halves meet through one shuffle.
    for (int cc = tid >> 1; cc < kc; cc += 128) {
        const int jh = tid & 1;
        const int j0 = jh ? (T + 1) / 2 : 0, j1 = jh ? T : (T + 1) / 2;
        const float* vp = a.v + (size_t)(h * kc + cc) * a.ld + base;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
        int j = j0;
        for (; j + 3 < j1; j += 4) {
            o0 += P[j] * vp[j]; o1 += P[j + 1] * vp[j + 1]; o2 += P[j + 2] * vp[j + 2]; o3 += P[j + 3] * vp[j + 3];
        }
        for (; j < j1; j++) o0 += P[j] * vp[j];
        float o = (o0 + o1) + (o2 + o3);
        o += __shfl_xor(o, 1, 64);
        if (jh == 0) {
            float orel = 0.f;
            if (a.win > 0) {
                for (int r = 0; r < a.px; r++) {
                    const int jj = i + r - a.win;
                    if (jj >= 0 && jj < T) orel += P[jj] * a.relv[(size_t)cc * a.px + r];
                }
            }
            a.o[(size_t)(h * kc + cc) * a.ld + base + i] = o + orel;
        }
    }
}
// ---------------------------------------------------------------------------------------------
// The one-query-per-workgroup attention with every global operand requested before the first dependent step (round 4).
// attention_kernel above walks a chain of dependent trips to L2 (q -> relK -> K -> ... -> V -> relV: ~16.8 us per launch at one
// 128-phoneme utterance, profiles/r04_c1_timeline.txt); here a wave requests its quarter of the channels of K AND V for all keys
// (2 x 24 x JPL registers per lane, lanes along the keys: 256-byte segments), its quarter of relK and the output stage's relV
// right after the query -- which one lane per channel loads and v_readlane hands to the wave as scalars, so no barrier (and no
// LDS round trip) separates it from the score product.  The scores are formed as above (four running sums over channels c % 4 per wave, partials
// added as (0 + 1) + (2 + 3): bit-identical when kc is a multiple of 16); P.V is summed per lane over its JPL keys, the 64 lane partials of a channel meet through LDS.
//   T <= 64 JPL keys, kc <= 96, px <= 16 (attention() falls back to the kernel above otherwise).
constexpr int ATR_CW = 24;      // channels per wave
constexpr int ATR_LD = 65;      // lane stride of the partial-output buffer (odd: conflict-free in both directions)

template <int JPL>
__global__ __launch_bounds__(256) void attention_reg_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int TMAX = 64 * JPL;
    const int b = blockIdx.z, h = blockIdx.y, i = blockIdx.x;
    const int T = seg_len(a.seg, b);
    if (i >= T) return;
    const size_t base = (size_t)seg_start(a.seg, b);
    const int kc = a.kc, px = a.px, win = a.win, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* red = sm;                      // [8]
    float* P = sm + 8;                    // [TMAX]
    float* part = P + TMAX;               // [4][TMAX] partial scores
    float* qr = part + 4 * TMAX;          // [4][32] partial relative-key logits
    float* buf = qr + 128;                // [4 * ATR_CW][ATR_LD] per-lane partial outputs
    const int cw = (kc + 3) / 4, c0 = wave * cw, nc = (c0 + cw < kc ? c0 + cw : kc) - c0;   // wave w owns channels [c0, c0 + nc)
    const float sq = sqrtf((float)kc);

    // ---- 1. every global operand, in the order of use.  Addresses are clamped into the tensors and the values masked afterwards:
    // no branch (and so no wait) sits between two requests
    float qv[ATR_CW], kv[ATR_CW][JPL], rk[ATR_CW], vv[ATR_CW][JPL], rv[16];
    int jc[JPL];
#pragma unroll
    for (int jj = 0; jj < JPL; jj++) jc[jj] = lane + 64 * jj < T ? lane + 64 * jj : T - 1;
    const int occ = tid >> 1, ohf = tid & 1;     // output stage: thread (channel, half of the lanes)
    const int occ_c = occ < kc ? occ : kc - 1, lane_px = lane < px ? lane : px - 1;
    // the query: lane c holds channel c0 + c; v_readlane hands the values to all lanes below (no LDS round trip, no barrier)
    const int qch = c0 + lane < kc ? c0 + lane : kc - 1;
    const float qlane = a.q[(size_t)(h * kc + qch) * a.ld + base + i];
#pragma unroll
    for (int c = 0; c < ATR_CW; c++) {
        const int ch = c0 + c < kc ? c0 + c : kc - 1;
#pragma unroll
        for (int jj = 0; jj < JPL; jj++) kv[c][jj] = a.k[(size_t)(h * kc + ch) * a.ld + base + jc[jj]];
    }
    if (win > 0) {
#pragma unroll
        for (int c = 0; c < ATR_CW; c++) {
            const int ch = c0 + c < kc ? c0 + c : kc - 1;
            rk[c] = a.relk[(size_t)ch * px + lane_px];
        }
    }
#pragma unroll
    for (int c = 0; c < ATR_CW; c++) {
        const int ch = c0 + c < kc ? c0 + c : kc - 1;
#pragma unroll
        for (int jj = 0; jj < JPL; jj++) vv[c][jj] = a.v[(size_t)(h * kc + ch) * a.ld + base + jc[jj]];
    }
    if (win > 0) {
#pragma unroll
        for (int r = 0; r < 16; r++) rv[r] = a.relv[(size_t)occ_c * px + (r < px ? r : px - 1)];
    }
    const float qscaled = qlane / sq;
#pragma unroll
    for (int c = 0; c < ATR_CW; c++) {
        const bool live = c < nc;
        qv[c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, qscaled), c));   // (past nc: a finite value against zeroed k / relK)
        rk[c] = (live && win > 0) ? rk[c] : 0.f;
#pragma unroll
        for (int jj = 0; jj < JPL; jj++) {
            const bool ok = live && lane + 64 * jj < T;
            kv[c][jj] = ok ? kv[c][jj] : 0.f;
            vv[c][jj] = ok ? vv[c][jj] : 0.f;
        }
    }

    // ---- 2. partial scores and partial relative-key logits of the wave's channels
#pragma unroll
    for (int jj = 0; jj < JPL; jj++) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int c = 0; c < ATR_CW; c += 4) {
            s0 += qv[c] * kv[c][jj]; s1 += qv[c + 1] * kv[c + 1][jj]; s2 += qv[c + 2] * kv[c + 2][jj]; s3 += qv[c + 3] * kv[c + 3][jj];   // (channels past nc: 0 * 0)
        }
        part[wave * TMAX + lane + 64 * jj] = (s0 + s1) + (s2 + s3);
    }
    if (win > 0 && lane < 32) {
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < ATR_CW; c++) s += qv[c] * rk[c];
        qr[wave * 32 + lane] = s;
    }
    __syncthreads();

    // ---- 3. softmax without a max shift (nn_softmax.cpp:5-28), one key per thread
    float e = 0.f;
    if (tid < T) {
        float sc = (part[tid] + part[TMAX + tid]) + (part[2 * TMAX + tid] + part[3 * TMAX + tid]);
        const int r = tid - i + win;
        if (win > 0 && r >= 0 && r < px) sc += (qr[r] + qr[32 + r]) + (qr[64 + r] + qr[96 + r]);
        e = expf(sc);
    }
    const float ws = wave_sum(e);
    if (lane == 0) red[wave] = ws;
    __syncthreads();
    const float sum = red[0] + red[1] + red[2] + red[3];
    if (tid < TMAX) P[tid] = tid < T ? e / sum : 0.f;
    __syncthreads();

    // ---- 4. P.V over the lane's keys
    float pj[JPL];
#pragma unroll
    for (int jj = 0; jj < JPL; jj++) pj[jj] = P[lane + 64 * jj];
#pragma unroll
    for (int c = 0; c < ATR_CW; c++) {
        float o = 0.f;
#pragma unroll
        for (int jj = 0; jj < JPL; jj++) o += pj[jj] * vv[c][jj];
        buf[(wave * ATR_CW + c) * ATR_LD + lane] = o;            // (rows past nc: zeros nobody reads)
    }
    __syncthreads();

    // ---- 5. the 64 lane partials of a channel: two threads x 32, then the banded relative-value term
    if (occ < kc) {
        const int wv = occ / cw, cl = occ - wv * cw;
        const float* p = buf + (wv * ATR_CW + cl) * ATR_LD + ohf * 32;
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
#pragma unroll
        for (int m = 0; m < 32; m += 4) { o0 += p[m]; o1 += p[m + 1]; o2 += p[m + 2]; o3 += p[m + 3]; }
        float o = (o0 + o1) + (o2 + o3);
        o += __shfl_xor(o, 1, 64);
        if (ohf == 0) {
            float orel = 0.f;
            if (win > 0) {
#pragma unroll
                for (int r = 0; r < 16; r++) {           // branch-free: absent terms add P * 0
                    const int jj = i + r - win;
                    const bool ok = r < px && jj >= 0 && jj < T;
                    orel += P[ok ? jj : 0] * (ok ? rv[r] : 0.f);
                }
            }
            a.o[(size_t)(h * kc + occ) * a.ld + base + i] = o + orel;
        }
    }
}
static size_t attention_reg_lds(int jpl) { return (size_t)(8 + 5 * 64 * jpl + 128 + 4 * ATR_CW * ATR_LD) * sizeof(float); }

// ---------------------------------------------------------------------------------------------
// The same attention on the matrix cores, 16 queries per workgroup (round 2).  The kernel above re-reads an utterance's
// K and V once per query (2 x kc x T floats per workgroup): fine for one short utterance, but 46 % of the text encoder's
// time at batch 8 (profiles/r02_b8_kernel_stats.csv: 98 us per launch).  Here a workgroup (4 waves) owns a block of 16
// queries of one head:
//   S^T tiles [16 keys x 16 queries] = K^T q on v_mfma_f32_16x16x4_f32, one 16-key tile per wave at a time, K fragments
//     straight from global memory (lanes run along the keys: 64-byte segments), q fragments from LDS;
//   + banded relative-key logits, exp (no max shift, nn_softmax.cpp:5-28), row sums through shuffles + LDS, P normalised
//     in LDS in [key][query] order;
//   O = P V: keys are the reduction axis, so V is staged through LDS in 64-key chunks ([key][channel], padded against bank
//     conflicts), every wave takes a quarter of each chunk's key steps for all kc / 16 channel tiles, the four partial
//     accumulators meet through LDS; the banded relative-value term is added per output element.
// K and V are read once per 16 queries instead of once per query.
typedef float attn_f32x4 __attribute__((ext_vector_type(4)));
constexpr int ATT_VLD_PAD = 16;     // vbuf row = kc + 16 floats: quads of a half-wave land in distinct banks

__global__ __launch_bounds__(256) void attention_mfma_kernel(AttnArgs a, int Tpad) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * 16;
    const int T = seg_len(a.seg, b);
    if (i0 >= T) return;
    const size_t base = (size_t)seg_start(a.seg, b);
    const int kc = a.kc, px = a.px, win = a.win;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, quad = lane >> 4;
    const int VLD = kc + ATT_VLD_PAD;
    float* qs = sm;                          // [kc][16]   q / sqrt(kc), zero for queries past the end
    float* qrel = qs + kc * 16;              // [16][32]   q . relK[:, r]
    float* rsum = qrel + 16 * 32;            // [4][16]
    float* P = rsum + 64;                    // [Tpad][16] exp(S) then softmax, [key][query]
    float* vbuf = P + (size_t)Tpad * 16;     // [64][VLD]  one V chunk; later the partial outputs [4][kc/16][64][4]
    const float sq = sqrtf((float)kc);
    {   // all of a thread's q loads are issued together (kc <= 128: at most 8 per thread)
        float qv[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int e = tid + u * 256, c = e >> 4, i = e & 15;
            qv[u] = (e < kc * 16 && i0 + i < T) ? a.q[(size_t)(h * kc + c) * a.ld + base + i0 + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int e = tid + u * 256; if (e < kc * 16) qs[e] = qv[u] / sq; }
    }
    __syncthreads();
    if (win > 0)
        for (int e = tid; e < 16 * px; e += 256) {
            const int i = e & 15, r = e >> 4;
            float s = 0.f;
            for (int c0 = 0; c0 < kc; c0 += 16) {        // 16 independent table loads per round
                float rk[16];
#pragma unroll
                for (int u = 0; u < 16; u++) rk[u] = a.relk[(size_t)(c0 + u) * px + r];
#pragma unroll
                for (int u = 0; u < 16; u++) s += qs[(c0 + u) * 16 + i] * rk[u];
            }
            qrel[i * 32 + r] = s;
        }
    __syncthreads();

    // ---- scores: wave w takes key tiles w, w + 4, ...
    float part[4] = {0.f, 0.f, 0.f, 0.f};    // row-sum partials of queries 4 quad + r (over this wave's tiles, this lane's key)
    const int ntile = (T + 15) / 16;
    for (int jt = wave; jt < ntile; jt += 4) {
        const int j = jt * 16 + col;         // this lane's key
        const bool kv = j < T;
        attn_f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const float* kp = a.k + (size_t)(h * kc) * a.ld + base + (kv ? j : 0);
        {   // the tile's whole K fragment (kc / 4 <= 32 k-steps of 4 channels) is requested at once
            float kb[32];
#pragma unroll
            for (int u = 0; u < 32; u++) { const int c = 4 * u + quad; kb[u] = (kv && c < kc) ? kp[(size_t)c * a.ld] : 0.f; }
#pragma unroll
            for (int u = 0; u < 32; u++) {
                const int c = 4 * u + quad;
                if (4 * u < kc) {
                    const float qa = qs[c * 16 + col];     // A[i = col][k = quad] = q of query col, channel c
                    // D[row = 4 quad + r][col]: rows = A's i index = QUERY, cols = B's j index = KEY
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qa, kb[u], acc, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int i = 4 * quad + r;                // query of this accumulator row; key = j (this lane's column)
            float sc = acc[r];
            const int rr = j - (i0 + i) + win;
            if (win > 0 && rr >= 0 && rr < px) sc += qrel[i * 32 + rr];
            const float e = (kv && i0 + i < T) ? expf(sc) : 0.f;
            P[(size_t)j * 16 + i] = e;                 // j < Tpad always
            part[r] += e;
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {                      // sum over the 16 keys held by the lanes of this quad
        float v = part[r];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        if (col == 0) rsum[wave * 16 + 4 * quad + r] = v;
    }
    __syncthreads();
    for (int e = tid; e < ntile * 256; e += 256) {     // P = exp / sum   (nn_softmax.cpp: divide every element by the row sum)
        const int i = e & 15;
        const float sum = (rsum[i] + rsum[16 + i]) + (rsum[32 + i] + rsum[48 + i]);
        P[e] = P[e] / sum;
    }
    __syncthreads();

    // ---- O = P V: 64-key chunks of V through LDS; wave w takes k-steps w, w + 4, ... of every chunk
    const int nct = kc / 16;                           // channel tiles (<= 8)
    attn_f32x4 oacc[8];
#pragma unroll
    for (int n = 0; n < 8; n++) oacc[n] = attn_f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j0 = 0; j0 < T; j0 += 64) {
        // stage V[kc][64 keys] -> vbuf[key][channel]; a thread reads runs of keys (coalesced) for one channel at a time
        for (int e0 = tid; e0 < kc * 64; e0 += 256 * 8) {      // 8 loads in flight per thread
            float vv[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = e0 + u * 256, c = e >> 6, jj = e & 63;
                vv[u] = (e < kc * 64 && j0 + jj < T) ? a.v[(size_t)(h * kc + c) * a.ld + base + j0 + jj] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int e = e0 + u * 256, c = e >> 6, jj = e & 63;
                if (e < kc * 64) vbuf[jj * VLD + c] = vv[u];
            }
        }
        __syncthreads();
        for (int ks = wave; ks < 16; ks += 4) {
            const int jj = 4 * ks + quad;              // key inside the chunk
            if (j0 + 4 * ks >= T) break;
            const float pa = P[(size_t)(j0 + jj) * 16 + col];          // A[i = col][k = quad] = P[query col][key jj]  (zero past T)
#pragma unroll
            for (int n = 0; n < 8; n++)
                if (n < nct) oacc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa, vbuf[jj * VLD + n * 16 + col], oacc[n], 0, 0, 0);
        }
        __syncthreads();
    }
    // ---- combine the four waves' partial outputs (vbuf is free now): red[wave][n][lane][r]
    float* red = vbuf;
#pragma unroll
    for (int n = 0; n < 8; n++)
        if (n < nct) {
#pragma unroll
            for (int r = 0; r < 4; r++) red[((wave * 8 + n) * 64 + lane) * 4 + r] = oacc[n][r];
        }
    __syncthreads();
    // D[row = 4 quad + r -> query][col -> channel n 16 + col]; one output element per (n, lane, r)
    for (int e = tid; e < nct * 256; e += 256) {
        const int n = e >> 8, ln = (e >> 2) & 63, r = e & 3;
        const int i = 4 * (ln >> 4) + r, ch = n * 16 + (ln & 15);
        if (i0 + i >= T) continue;
        float o = 0.f;
#pragma unroll
        for (int w = 0; w < 4; w++) o += red[((w * 8 + n) * 64 + ln) * 4 + r];
        float orel = 0.f;
        if (win > 0) {
            float rv[32];
#pragma unroll
            for (int rr = 0; rr < 32; rr++) rv[rr] = rr < px ? a.relv[(size_t)ch * px + rr] : 0.f;
#pragma unroll
            for (int rr = 0; rr < 32; rr++) {
                const int jj = i0 + i + rr - win;
                if (rr < px && jj >= 0 && jj < T) orel += P[(size_t)jj * 16 + i] * rv[rr];
            }
        }
        a.o[(size_t)(h * kc + ch) * a.ld + base + i0 + i] = o + orel;
    }
}

static size_t attention_mfma_lds(const AttnArgs& a, int Tpad) {
    const size_t red = (size_t)4 * 8 * 64 * 4, vb = (size_t)64 * (a.kc + ATT_VLD_PAD);
    return ((size_t)a.kc * 16 + 16 * 32 + 64 + (size_t)Tpad * 16 + (vb > red ? vb : red)) * sizeof(float);
}

void attention(const AttnArgs& a, hipStream_t st) {
    if (a.max_len <= 0 || a.B <= 0) return;
    {   // matrix-core form: 16 queries per workgroup, whenever its LDS footprint fits and the model shape is covered
        static const bool no_mfma = exp_flag("STS_NO_ATTN_MFMA");   // experiment knob
        const int Tpad = (a.max_len + 63) / 64 * 64;
        const size_t lds = attention_mfma_lds(a, Tpad);
        // Measured (docs/HISTORY.md 5b): the block kernel is one long dependent chain per workgroup -- with 16 workgroups (one
        // 128-phoneme utterance) it takes 26 us against 11.5 us for the one-query-per-workgroup kernel; from about a
        // hundred workgroups on it wins (batch 8: 98 -> 45 us per launch, text encoder 1.21 -> 0.88 ms)
        const long wgs = (long)((a.max_len + 15) / 16) * a.nheads * a.B;
        const long min_wgs = a.block_min_wgs > 0 ? a.block_min_wgs : 96;
        if (!no_mfma && wgs >= min_wgs && a.kc % 16 == 0 && a.kc <= 128 && a.px <= 32 && lds <= 150 * 1024) {
            if (lds > 48 * 1024)
                hipFuncSetAttribute((const void*)attention_mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(attention_mfma_kernel, dim3((a.max_len + 15) / 16, a.nheads, a.B), dim3(256), lds, st, a, Tpad);
            return;
        }
    }
    if (a.attn_reg && a.kc <= 4 * ATR_CW && a.px <= 16 && a.max_len <= 256) {   // operands in registers (round 4)
        if (a.max_len <= 128) hipLaunchKernelGGL(attention_reg_kernel<2>, dim3(a.max_len, a.nheads, a.B), dim3(256), attention_reg_lds(2), st, a);
        else hipLaunchKernelGGL(attention_reg_kernel<4>, dim3(a.max_len, a.nheads, a.B), dim3(256), attention_reg_lds(4), st, a);
        return;
    }
    size_t lds = (size_t)(a.kc + 8 + (a.px > 16 ? a.px : 16) + 5 * (size_t)a.max_len) * sizeof(float);
    if (lds > 48 * 1024)
        hipFuncSetAttribute((const void*)attention_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(attention_kernel, dim3(a.max_len, a.nheads, a.B), dim3(256), lds, st, a);
}

// ---------------------------------------------------------------------------------------------
__global__ void add_ubias_kernel(float* y, long ld, const float* u, SegView seg, int B) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int pos = blockIdx.x * 256 + threadIdx.x;
    if (pos >= seg_len(seg, b)) return;
    y[(size_t)c * ld + seg_start(seg, b) + pos] += u[(size_t)c * B + b];
}
void add_ubias(float* y, long ld, const float* u, int C, SegView seg, int B, int max_len, hipStream_t st) {
    if (max_len <= 0 || C <= 0 || B <= 0) return;
    hipLaunchKernelGGL(add_ubias_kernel, dim3((max_len + 255) / 256, C, B), dim3(256), 0, st, y, ld, u, seg, B);
}

// g[c][b] = emb_g(sid_b, c) = ptr[c*spk_num + sid]  (/root/reference/src/models/SynthesizerTrn.cpp:159,364-372)
__global__ void gather_speaker_kernel(const float* emb_g, int spk_num, int gin, const int* sid, int B, float* g) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= gin * B) return;
    int c = i / B, b = i - c * B;
    int s = sid[b];
    if (s < 0 || s >= spk_num) s = 0;
    g[i] = emb_g[(size_t)c * spk_num + s];
}
void gather_speaker(const float* emb_g, int spk_num, int gin, const int* sid, int B, float* g, hipStream_t st) {
    int n = gin * B;
    if (n <= 0) return;
    hipLaunchKernelGGL(gather_speaker_kernel, dim3((n + 255) / 256), dim3(256), 0, st, emb_g, spk_num, gin, sid, B, g);
}

struct SumPtrs { const float* r[8]; };
__global__ void sum_scale_kernel(float* y, SumPtrs p, int count, float div, long n, int vec) {
    long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    if (vec && i + 3 < n) {
        float4 a = *reinterpret_cast<const float4*>(p.r[0] + i);
        for (int k = 1; k < count; k++) {
            const float4 b = *reinterpret_cast<const float4*>(p.r[k] + i);
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        a.x = a.x / div; a.y = a.y / div; a.z = a.z / div; a.w = a.w / div;
        *reinterpret_cast<float4*>(y + i) = a;
    } else {
        for (long j = i; j < n && j < i + 4; j++) {
            float a = p.r[0][j];
            for (int k = 1; k < count; k++) a += p.r[k][j];
            y[j] = a / div;
        }
    }
}
void sum_scale(float* y, const float* const* r, int count, long n, hipStream_t st) {
    if (n <= 0 || count <= 0 || count > 8) return;
    SumPtrs p;
    for (int k = 0; k < 8; k++) p.r[k] = r[k < count ? k : 0];
    int vec = ((uintptr_t)y & 15) == 0;
    for (int k = 0; k < count; k++) vec = vec && (((uintptr_t)r[k] & 15) == 0);
    long thr = (n + 3) / 4;
    hipLaunchKernelGGL(sum_scale_kernel, dim3((unsigned)((thr + 255) / 256)), dim3(256), 0, st, y, p, count, (float)count, n, vec);
}

// channel reversal (/root/reference/src/nn_op/nn_flip.cpp:3-15); only needed when n_flows is odd --
// otherwise the flips are folded into the coupling weights at load time.
__global__ void flip_copy_kernel(const float* x, long ld, int C, long n, float* tmp) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = blockIdx.y;
    tmp[(size_t)c * n + i] = x[(size_t)(C - 1 - c) * ld + i];
}
__global__ void copy_back_kernel(float* x, long ld, long n, const float* tmp) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = blockIdx.y;
    x[(size_t)c * ld + i] = tmp[(size_t)c * n + i];
}
void flip_channels(float* x, long ld, int C, long n, float* tmp, hipStream_t st) {
    if (n <= 0 || C <= 0) return;
    dim3 g((unsigned)((n + 255) / 256), C);
    hipLaunchKernelGGL(flip_copy_kernel, g, dim3(256), 0, st, x, ld, C, n, tmp);
    hipLaunchKernelGGL(copy_back_kernel, g, dim3(256), 0, st, x, ld, n, tmp);
}

// ---------------------------------------------------------------------------------------------
// (the inverse rational-quadratic spline itself: devmath.hpp rq_spline_inverse)
// one reverse ConvFlow step of the stochastic duration predictor including the channel flip
// (/root/reference/src/models/StochasticDurationPredictor.cpp:136-142): (r0, r1) -> (spline(r1 | h), r0)
__global__ void spline_step_kernel(const float* h, long ld, float fs, const float* r0, const float* r1,
                                   float* o0, float* o1, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float hv[29];
#pragma unroll
    for (int j = 0; j < 29; j++) hv[j] = h[(size_t)j * ld + i];
    const float a = r0 ? r0[i] : 0.f;             // null inputs = the all-zero latent of the first flow (noise scale 0)
    o0[i] = rq_spline_inverse(r1 ? r1[i] : 0.f, hv, fs);
    o1[i] = a;
}
void spline_step(const float* h, long ld, float filter_sqrt, const float* r0, const float* r1, float* o0, float* o1,
                 long n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(spline_step_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, h, ld, filter_sqrt, r0, r1, o0, o1, n);
}

// ---------------------------------------------------------------------------------------------
// durations: w = exp(logw) * lengthScale ; ceil ; (int) ; inclusive scan ; frames = max(sum, 1)
// (/root/reference/src/models/SynthesizerTrn.cpp:376-378, 304-321; ElementwiseAffine.cpp:44-58)
constexpr int kMaxDur = 100000;   // sanity clamp per phoneme (the reference has none)
__global__ __launch_bounds__(256) void durations_kernel(const float* r0, int sdp, float ea_m, float ea_logs,
                                                        const float* ls, const int* forced, float* logw_out,
                                                        int* dur, int* cum, int* frames, SegView seg,
                                                        int* host_out, long total, int seq, unsigned* arrive, int B,
                                                        int* len_out, int* win_len_out, int cap) {
    __shared__ int wsum[4];
    __shared__ int carry_s;
    const int b = blockIdx.x;
    const int T = seg_len(seg, b);
    const size_t base = (size_t)seg_start(seg, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    const float scale = ls[b];
    for (int t0 = 0; t0 < T; t0 += 256) {
        const int t = t0 + tid;
        int d = 0;
        if (t < T) {
            float lw = r0[base + t];
            if (sdp) lw = (lw - ea_m) * expf(ea_logs * (-1.0f));
            if (logw_out) logw_out[base + t] = lw;
            if (forced) d = forced[base + t];
            else {
                const float w = expf(lw) * scale;
                const float c = ceilf(w);
                d = (c >= (float)kMaxDur) ? kMaxDur : (c > 0.f ? (int)c : 0);
            }
            if (d < 0) d = 0;
            dur[base + t] = d;
            if (host_out) host_out[1 + base + t] = d;
        }
        // inclusive scan inside the wave, then across the 4 waves
        int v = d;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int u = __shfl_up(v, o, 64); if (lane >= o) v += u; }
        if (lane == 63) wsum[wave] = v;
        __syncthreads();
        int pre = carry_s;
        for (int k = 0; k < wave; k++) pre += wsum[k];
        if (t < T) cum[base + t] = pre + v;
        __syncthreads();
        if (tid == 255) carry_s = pre + v;
        __syncthreads();
    }
    if (tid == 0) {
        const int f = carry_s < 1 ? 1 : carry_s;
        frames[b] = f;
        // a launch-ahead run: the kernels that follow read the utterance's length from these device words (never more than the
        // capacity they were launched for -- the host repeats the run if the count was larger)
        if (len_out) { const int fc = f < cap ? f : cap; len_out[b] = fc; win_len_out[b] = fc; }
    }
    if (host_out) {
        // publish to the host: results first (system scope), then one arrival per workgroup; the last arriver writes the flag
        if (tid == 0) host_out[1 + total + b] = carry_s < 1 ? 1 : carry_s;
        __threadfence_system();
        __syncthreads();
        if (tid == 0) {
            const unsigned prev = atomicAdd(arrive, 1u);
            if (prev == (unsigned)B - 1u) {
                *arrive = 0u;                                   // ready for the next run (same stream: ordered)
                __threadfence_system();
                __hip_atomic_store(host_out, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // word 0: never holds data
            }
        }
    }
}
void durations(const float* r0, int sdp, float ea_m, float ea_logs, const float* ls, const int* forced,
               float* logw_out, int* dur, int* cum, int* frames, SegView seg, int B, hipStream_t st,
               int* host_out, long total, int seq, unsigned* arrive, int* len_out, int* win_len_out, int cap) {
    if (B <= 0) return;
    hipLaunchKernelGGL(durations_kernel, dim3(B), dim3(256), 0, st, r0, sdp, ea_m, ea_logs, ls, forced, logw_out, dur, cum, frames, seg,
                       host_out, total, seq, arrive, B, len_out, win_len_out, cap);
}

// length regulator: frame f of utterance b copies phoneme i with cum[i-1] <= f < cum[i]
__global__ __launch_bounds__(256) void expand_frames_kernel(const float* m, long m_ld, const int* cum, SegView segT,
                                                            SegView segF, int C, float* z, long z_ld) {
    const int b = blockIdx.z;
    const int F = seg_len(segF, b), T = seg_len(segT, b);
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= F) return;
    const size_t tb = (size_t)seg_start(segT, b), fb = (size_t)seg_start(segF, b);
    // the binary search is a chain of ~log2(T) dependent reads: through LDS (one trip to L2 for the whole table) instead of ~7 trips
    __shared__ int cs[1024];
    const bool staged = T <= 1024;
    if (staged) {
        for (int i = threadIdx.x; i < T; i += 256) cs[i] = cum[tb + i];
        __syncthreads();
    }
    if (f >= F) return;
    int lo = 0, hi = T;   // first i with cum[i] > f
    if (staged) { while (lo < hi) { int mid = (lo + hi) >> 1; if (cs[mid] > f) hi = mid; else lo = mid + 1; } }
    else { while (lo < hi) { int mid = (lo + hi) >> 1; if (cum[tb + mid] > f) hi = mid; else lo = mid + 1; } }
    const int c0 = blockIdx.y * 16, c1 = c0 + 16 < C ? c0 + 16 : C;
    if (lo >= T) { for (int c = c0; c < c1; c++) z[(size_t)c * z_ld + fb + f] = 0.f; return; }
#pragma unroll 4
    for (int c = c0; c < c1; c++) z[(size_t)c * z_ld + fb + f] = m[(size_t)c * m_ld + tb + lo];
}
void expand_frames(const float* m, long m_ld, const int* cum, SegView segT, SegView segF, int C, float* z, long z_ld,
                   int B, int max_frames, hipStream_t st) {
    if (B <= 0 || max_frames <= 0) return;
    hipLaunchKernelGGL(expand_frames_kernel, dim3((max_frames + 255) / 256, (C + 15) / 16, B), dim3(256), 0, st, m, m_ld, cum,
                       segT, segF, C, z, z_ld);
}

// ---------------------------------------------------------------------------------------------
// MB-iSTFT tail.  Spectrum: X_k = exp(logmag_k) * e^{i * pi * sin(phase_k)}, k = 0..8
// (/root/reference/src/models/Generator_MBB.cpp:186-197, modules/iStft.cpp:63-75)
__global__ void istft_spectrum_kernel(const float* sb, long ld, int rows, float* spec, long total) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int band = blockIdx.y / 9, k = blockIdx.y - band * 9;
    const float mag = expf(sb[(size_t)(band * 18 + k) * ld + i]);
    const float ph = sinf(sb[(size_t)(band * 18 + 9 + k) * ld + i]) * 3.14159265358979323846f;
    spec[(size_t)(band * 18 + k) * ld + i] = cosf(ph) * mag;
    spec[(size_t)(band * 18 + 9 + k) * ld + i] = sinf(ph) * mag;
}
void istft_spectrum(const float* sb, long ld, int rows, float* spec, long total, hipStream_t st) {
    if (total <= 0) return;
    hipLaunchKernelGGL(istft_spectrum_kernel, dim3((unsigned)((total + 255) / 256), rows / 2), dim3(256), 0, st, sb, ld, rows, spec, total);
}

// 16-point real inverse DFT (bins 0..8; Im(DC) = Im(Nyquist) = 0; scale 1/16 -- Eigen's kissfft real
// inverse, unsupported/Eigen/src/FFT/ei_kissfft_impl.h:374-405), Hann window, overlap-add hop 4,
// divide by the overlap-added squared window where > 1e-14, crop 8 (modules/iStft.cpp:46-124, hann.cpp:3-9)
__constant__ float kHann[16] = {0.0f, 0.03806023f, 0.14644661f, 0.30865828f, 0.5f, 0.69134172f, 0.85355339f, 0.96193977f,
                                1.0f, 0.96193977f, 0.85355339f, 0.69134172f, 0.5f, 0.30865828f, 0.14644661f, 0.03806023f};
__constant__ float kHannPow[16] = {0.0f, 0.00144858f, 0.02144661f, 0.09526994f, 0.25f, 0.47795337f, 0.72855339f, 0.92532811f,
                                   1.0f, 0.92532811f, 0.72855339f, 0.47795337f, 0.25f, 0.09526994f, 0.02144661f, 0.00144858f};
__constant__ float kCos16[16] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f, 0.0f,
                                 -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f, -1.0f,
                                 -0.92387953251128674f, -0.70710678118654752f, -0.38268343236508977f, 0.0f,
                                 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f};
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* spec, long ld, int band_rows, SegView segf,
                                                        float* tm, long tm_ld, SegView segt) {
    const int b = blockIdx.z, band = blockIdx.y;
    const int frames = seg_len(segf, b);
    const int n = (frames - 1) * 4;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= n) return;
    const size_t fb = (size_t)seg_start(segf, b), tb = (size_t)seg_start(segt, b);
    const float* re = spec + (size_t)(band * band_rows) * ld + fb;
    const float* im = re + (size_t)9 * ld;
    const int p = s + 8;
    int f0 = (p - 15 + 3) >> 2; if (f0 < 0) f0 = 0;      // ceil((p-15)/4)
    int f1 = p >> 2; if (f1 > frames - 1) f1 = frames - 1;
    float acc = 0.f, ws = 0.f;
    for (int f = f0; f <= f1; f++) {
        const int idx = p - 4 * f;
        float v = re[f] + ((idx & 1) ? -re[(size_t)8 * ld + f] : re[(size_t)8 * ld + f]);
#pragma unroll
        for (int k = 1; k < 8; k++) {
            const int ang = (k * idx) & 15;
            v += 2.0f * (re[(size_t)k * ld + f] * kCos16[ang] - im[(size_t)k * ld + f] * kCos16[(ang + 12) & 15]);
        }
        acc += (v * 0.0625f) * kHann[idx];
        ws += kHannPow[idx];
    }
    if (ws > 1e-14f) acc = acc / ws;
    tm[(size_t)band * tm_ld + tb + s] = acc;
}
void istft_ola(const float* spec, long ld, int bands, int band_rows, SegView seg_frames, float* tm, long tm_ld,
               SegView seg_tm, int B, int max_n, hipStream_t st) {
    if (B <= 0 || max_n <= 0) return;
    hipLaunchKernelGGL(istft_ola_kernel, dim3((max_n + 255) / 256, bands, B), dim3(256), 0, st, spec, ld, band_rows,
                       seg_frames, tm, tm_ld, seg_tm);
}

// x4 zero-stuff (gain) + 63-tap synthesis FIR, polyphase: only every 4th tap meets a non-zero sample.
// PQMF: /root/reference/src/modules/pqmf.cpp:104-115 ; MS learned filter: models/Generator_MS.cpp:225-226
__global__ __launch_bounds__(256) void synth_fir_kernel(const float* tm, long tm_ld, SegView segt, const float* fir,
                                                        int ntap, int pad, float gain, float bias, float* wave, int16_t* pcm,
                                                        SegView sego) {
    const int b = blockIdx.y;
    const int n = seg_len(segt, b), N = 4 * n;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const size_t tb = (size_t)seg_start(segt, b), ob = (size_t)seg_start(sego, b);
    int tau0 = ((pad - i) % 4 + 4) % 4;   // first tau with (i + tau - pad) % 4 == 0
    float s = 0.f;
    for (int tau = tau0; tau < ntap; tau += 4) {
        const int u = i + tau - pad;
        if (u < 0 || u >= N) continue;
        const int t = u >> 2;
#pragma unroll
        for (int q = 0; q < 4; q++) s += (gain * tm[(size_t)q * tm_ld + tb + t]) * fir[tau * 4 + q];
    }
    s += bias;                 // (MS: the learned conv's bias, when the blob carries one; 0 otherwise)
    if (wave) wave[ob + i] = s;
    pcm[ob + i] = pcm_cast(s);
}
void synth_fir(const float* tm, long tm_ld, SegView seg_tm, const float* fir, int ntap, int pad, float gain, float bias, float* wave,
               int16_t* pcm, SegView seg_out, int B, int max_n, hipStream_t st) {
    if (B <= 0 || max_n <= 0) return;
    hipLaunchKernelGGL(synth_fir_kernel, dim3((4 * max_n + 255) / 256, B), dim3(256), 0, st, tm, tm_ld, seg_tm, fir, ntap,
                       pad, gain, bias, wave, pcm, seg_out);
}

// ---- the three launches above as ONE (round 6; SURVEY.md 7 planned the tail as one kernel): a workgroup owns 1024 output samples of one
// utterance = 256 time-domain samples per band (+ 8 either side for the synthesis filter) = 71 iSTFT frames.  Phase A: the spectrum of those
// frames (exp / pi sin / cos, sin) into LDS; phase B: 16-point inverse DFT + Hann + overlap-add + normalisation per band into LDS; phase C: the x4
// zero-stuffed polyphase synthesis filter (PQMF bank or the MS model's learned filter) + the int16 cast.  The spectrum (72 floats per frame) and the
// band signals (16 floats per frame) never reach memory: 0.76 -> ~0.3 ms of a 33 ms step at 64 MB-iSTFT utterances, three ~8 us launches -> one at
// one utterance.  Same expressions in the same order as the three kernels above (which remain for the plain-iSTFT decoder and as the lab A/B).
// /root/reference/src/models/Generator_MBB.cpp:186-203, modules/iStft.cpp:46-124, modules/pqmf.cpp:104-115
constexpr int TF_T = 256, TF_HALO = 8, TF_TW = TF_T + 2 * TF_HALO, TF_FR = 72;     // band samples per workgroup, halo, staged width, frames staged (71 used)
__global__ __launch_bounds__(256) void istft_tail_fused_kernel(const float* sb, long ld, SegView segf, const float* fir, int ntap, int pad, float gain,
                                                               float bias, float* wave, int16_t* pcm, SegView sego) {
    __shared__ float spec[4][18][TF_FR];
    __shared__ float tmv[4][TF_TW];
    __shared__ float firs[256];
    const int b = blockIdx.y;
    const int frames = seg_len(segf, b);
    const int n = (frames - 1) * 4, N = 4 * n;
    const int t0 = blockIdx.x * TF_T;
    if (t0 >= n) return;
    const size_t fb = (size_t)seg_start(segf, b), ob = (size_t)seg_start(sego, b);
    const int tid = threadIdx.x;
    const int fl0 = (t0 >> 2) - 3;                        // first staged frame (t0 is a multiple of 4)
    if (tid < ntap * 4 && tid < 256) firs[tid] = fir[tid];
    // ---- A: spectrum of frames [fl0, fl0 + 71)
    for (int it = tid; it < 4 * 9 * TF_FR; it += 256) {
        const int fi = it % TF_FR, kk = it / TF_FR;       // kk = band * 9 + k
        const int band = kk / 9, k = kk - band * 9;
        const int f = fl0 + fi;
        if (f >= 0 && f < frames) {
            const float mag = expf(sb[(size_t)(band * 18 + k) * ld + fb + f]);
            const float ph = sinf(sb[(size_t)(band * 18 + 9 + k) * ld + fb + f]) * 3.14159265358979323846f;
            spec[band][k][fi] = cosf(ph) * mag;
            spec[band][9 + k][fi] = sinf(ph) * mag;
        }
    }
    __syncthreads();
    // ---- B: band samples [t0 - 8, t0 + 264)
    for (int it = tid; it < 4 * TF_TW; it += 256) {
        const int band = it / TF_TW, sl = it - band * TF_TW;
        const int s = t0 - TF_HALO + sl;
        float acc = 0.f;
        if (s >= 0 && s < n) {
            const int p = s + 8;
            int f0 = (p - 15 + 3) >> 2; if (f0 < 0) f0 = 0;
            int f1 = p >> 2; if (f1 > frames - 1) f1 = frames - 1;
            float ws = 0.f;
            for (int f = f0; f <= f1; f++) {
                const int idx = p - 4 * f, fi = f - fl0;
                float v = spec[band][0][fi] + ((idx & 1) ? -spec[band][8][fi] : spec[band][8][fi]);
#pragma unroll
                for (int k = 1; k < 8; k++) {
                    const int ang = (k * idx) & 15;
                    v += 2.0f * (spec[band][k][fi] * kCos16[ang] - spec[band][9 + k][fi] * kCos16[(ang + 12) & 15]);
                }
                acc += (v * 0.0625f) * kHann[idx];
                ws += kHannPow[idx];
            }
            if (ws > 1e-14f) acc = acc / ws;
        }
        tmv[band][sl] = acc;
    }
    __syncthreads();
    // ---- C: output samples [4 t0, 4 t0 + 1024)
    for (int j = tid; j < 4 * TF_T; j += 256) {
        const int i = 4 * t0 + j;
        if (i >= N) break;
        const int tau0 = ((pad - i) % 4 + 4) % 4;
        float s = 0.f;
        for (int tau = tau0; tau < ntap; tau += 4) {
            const int u = i + tau - pad;
            if (u < 0 || u >= N) continue;
            const int tl = (u >> 2) - t0 + TF_HALO;
#pragma unroll
            for (int q = 0; q < 4; q++) s += (gain * tmv[q][tl]) * firs[tau * 4 + q];
        }
        s += bias;
        if (wave) wave[ob + i] = s;
        pcm[ob + i] = pcm_cast(s);
    }
}
bool istft_tail_fused_ok(int bands, int ntap, int pad) { return bands == 4 && ntap * 4 <= 256 && pad <= 4 * TF_HALO - 1 && ntap - 1 - pad <= 4 * TF_HALO - 1; }
void istft_tail_fused(const float* sb, long ld, SegView seg_frames, const float* fir, int ntap, int pad, float gain, float bias, float* wave, int16_t* pcm,
                      SegView seg_out, int B, int max_n, hipStream_t st) {
    if (B <= 0 || max_n <= 0) return;      // max_n = band samples of the longest utterance
    hipLaunchKernelGGL(istft_tail_fused_kernel, dim3((max_n + TF_T - 1) / TF_T, B), dim3(256), 0, st, sb, ld, seg_frames, fir, ntap, pad, gain, bias, wave, pcm, seg_out);
}

__global__ void quantize_pcm_kernel(const float* wave, int16_t* pcm, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) pcm[i] = pcm_cast(wave[i]);
}
void quantize_pcm(const float* wave, int16_t* pcm, long n, hipStream_t st) {
    if (n <= 0) return;
    hipLaunchKernelGGL(quantize_pcm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, wave, pcm, n);
}

}  // namespace sts
