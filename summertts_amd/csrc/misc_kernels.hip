// misc_kernels.hip -- the non-convolution kernels of the acoustic path (all HBM/latency bound):
// embedding, LayerNorm, windowed relative-position attention, the duration-side spline flow,
// the length regulator, the MB-iSTFT synthesis tail and the int16 quantiser.  Reductions use
// wave64 shuffles; every kernel reads/writes time-contiguous rows so global accesses coalesce.
#include "kernels.hpp"
#include "devmath.hpp"

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
void attention(const AttnArgs& a, hipStream_t st) {
    if (a.max_len <= 0 || a.B <= 0) return;
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
// Inverse rational-quadratic spline, 10 bins, linear tails outside (-5, 5).
// /root/reference/src/modules/ConvFlow.cpp:80-240 (+ searchsorted :57-78, on the cumulative HEIGHTS
// because this is the inverse direction).  Everything lives in registers; one thread per time step.
__device__ float rq_spline_inverse(float x, const float* h, float filter_sqrt) {
    constexpr int NB = 10;
    const float tail = 5.0f;
    if (!(x < tail && x > -tail)) return x;
    float uw[NB], uh[NB];
    float sw = 0.f, sh = 0.f;
#pragma unroll
    for (int i = 0; i < NB; i++) {
        uw[i] = expf(h[i] / filter_sqrt); sw += uw[i];
        uh[i] = expf(h[NB + i] / filter_sqrt); sh += uh[i];
    }
    float cw[NB + 1], ch[NB + 1], der[NB + 1];
    float aw = 0.f, ah = 0.f;
    cw[0] = -tail; ch[0] = -tail;
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const float w = (uw[i] / sw) * (float)(1 - 1e-3 * NB) + (float)1e-3;
        const float hh = (uh[i] / sh) * (float)(1 - 1e-3 * NB) + (float)1e-3;
        aw += w; ah += hh;
        cw[i + 1] = aw * (tail - (-tail)) + (-tail);
        ch[i + 1] = ah * (tail - (-tail)) + (-tail);
    }
    cw[NB] = tail; ch[NB] = tail;
    der[0] = softplus_ref(0.5397424172369522f) + (float)1e-3;
    der[NB] = der[0];
#pragma unroll
    for (int i = 1; i < NB; i++) der[i] = softplus_ref(h[2 * NB + i - 1]) + (float)1e-3;
    int cnt = 0;
#pragma unroll
    for (int j = 0; j <= NB; j++) {
        float edge = ch[j];
        if (j == NB) edge = edge + 1e-6f;
        cnt += (x >= edge) ? 1 : 0;
    }
    int bi = cnt - 1;
    bi = bi < 0 ? 0 : (bi > NB - 1 ? NB - 1 : bi);   // the reference would assert out of range
    float in_cw = 0.f, in_w = 0.f, in_ch = 0.f, in_h = 0.f, d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int j = 0; j < NB; j++)
        if (j == bi) { in_cw = cw[j]; in_w = cw[j + 1] - cw[j]; in_ch = ch[j]; in_h = ch[j + 1] - ch[j]; d0 = der[j]; d1 = der[j + 1]; }
    const float delta = in_h / in_w;
    const float xm = x - in_ch;
    const float aa = xm * (d0 + d1 - delta * 2.0f) + in_h * (delta - d0);
    const float bq = in_h * d0 - xm * (d0 + d1 - 2.0f * delta);
    const float cc = -(delta * xm);
    const float disc = bq * bq - aa * cc * 4.0f;
    const float root = (cc * 2.0f) / (-bq - sqrtf(disc));
    return root * in_w + in_cw;
}

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
                                                        int* host_out, long total, int seq, unsigned* arrive, int B) {
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
    if (tid == 0) frames[b] = carry_s < 1 ? 1 : carry_s;
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
               int* host_out, long total, int seq, unsigned* arrive) {
    if (B <= 0) return;
    hipLaunchKernelGGL(durations_kernel, dim3(B), dim3(256), 0, st, r0, sdp, ea_m, ea_logs, ls, forced, logw_out, dur, cum, frames, seg,
                       host_out, total, seq, arrive, B);
}

// length regulator: frame f of utterance b copies phoneme i with cum[i-1] <= f < cum[i]
__global__ __launch_bounds__(256) void expand_frames_kernel(const float* m, long m_ld, const int* cum, SegView segT,
                                                            SegView segF, int C, float* z, long z_ld) {
    const int b = blockIdx.z;
    const int F = seg_len(segF, b), T = seg_len(segT, b);
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    const size_t tb = (size_t)seg_start(segT, b), fb = (size_t)seg_start(segF, b);
    int lo = 0, hi = T;   // first i with cum[i] > f
    while (lo < hi) { int mid = (lo + hi) >> 1; if (cum[tb + mid] > f) hi = mid; else lo = mid + 1; }
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
                                                        int ntap, int pad, float gain, float* wave, int16_t* pcm,
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
    if (wave) wave[ob + i] = s;
    pcm[ob + i] = pcm_cast(s);
}
void synth_fir(const float* tm, long tm_ld, SegView seg_tm, const float* fir, int ntap, int pad, float gain, float* wave,
               int16_t* pcm, SegView seg_out, int B, int max_n, hipStream_t st) {
    if (B <= 0 || max_n <= 0) return;
    hipLaunchKernelGGL(synth_fir_kernel, dim3((4 * max_n + 255) / 256, B), dim3(256), 0, st, tm, tm_ld, seg_tm, fir, ntap,
                       pad, gain, wave, pcm, seg_out);
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
