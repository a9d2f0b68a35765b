// devmath.hpp -- transcendental formulas that DEFINE parity with the reference.
// The reference does not call tanhf/erff: it composes everything from exp/log, and the exact
// composition matters at the 1e-7 level the parity tests check.  Accurate expf/logf are used
// (never the fast __expf approximations).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace sts {

// /root/reference/src/nn_op/nn_tanh.cpp:6-21: (e^x - e^-x) / (e^x + e^-x), inf -> 1e10, denominator floor 1e-8
__device__ __forceinline__ float tanh_ref(float x) {
    float e = expf(x), n = expf(-x);
    if (isinf(e)) e = 1e10f;
    if (isinf(n)) n = 1e10f;
    float m0 = e - n, m1 = e + n;
    if (m1 < 1e-8f) m1 = 1e-8f;
    return m0 / m1;
}
// /root/reference/src/nn_op/nn_sigmoid.cpp:3-7
__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }
// /root/reference/src/nn_op/nn_gelu.cpp:4-14 (tanh approximation through nn_tanh)
__device__ __forceinline__ float gelu_ref(float x) {
    float t = (x + (x * x * x) * 0.044715f) * 0.7978845608028654f;
    return (tanh_ref(t) + 1.0f) * x * 0.5f;
}
// /root/reference/src/nn_op/nn_softplus.cpp:3-8: log(e^x + 1), no threshold
__device__ __forceinline__ float softplus_ref(float x) { return logf(expf(x) + 1.0f); }

// Inverse rational-quadratic spline, 10 bins, linear tails outside (-5, 5).
// /root/reference/src/modules/ConvFlow.cpp:80-240 (+ searchsorted :57-78, on the cumulative HEIGHTS
// because this is the inverse direction).  Everything lives in registers; one thread per time step.
// Two parts so that a fused kernel can spread the 29 transcendental element transforms over threads (col_layer.hip):
//   rq_spline_param(j, h_j)   the per-parameter transform: exp(h / sqrt(filter)) for the 10 widths and 10 heights, softplus(h) + 1e-3
//                             for the 9 inner derivatives
//   rq_spline_inverse_t(x, t) the rest, on the 29 transformed values
__device__ __forceinline__ float rq_spline_param(int j, float h, float filter_sqrt) {
    return j < 20 ? expf(h / filter_sqrt) : softplus_ref(h) + (float)1e-3;
}
__device__ __forceinline__ float rq_spline_inverse_t(float x, const float* t) {
    constexpr int NB = 10;
    const float tail = 5.0f;
    if (!(x < tail && x > -tail)) return x;
    float sw = 0.f, sh = 0.f;
#pragma unroll
    for (int i = 0; i < NB; i++) { sw += t[i]; sh += t[NB + i]; }
    float cw[NB + 1], ch[NB + 1], der[NB + 1];
    float aw = 0.f, ah = 0.f;
    cw[0] = -tail; ch[0] = -tail;
#pragma unroll
    for (int i = 0; i < NB; i++) {
        const float w = (t[i] / sw) * (float)(1 - 1e-3 * NB) + (float)1e-3;
        const float hh = (t[NB + i] / sh) * (float)(1 - 1e-3 * NB) + (float)1e-3;
        aw += w; ah += hh;
        cw[i + 1] = aw * (tail - (-tail)) + (-tail);
        ch[i + 1] = ah * (tail - (-tail)) + (-tail);
    }
    cw[NB] = tail; ch[NB] = tail;
    der[0] = softplus_ref(0.5397424172369522f) + (float)1e-3;
    der[NB] = der[0];
#pragma unroll
    for (int i = 1; i < NB; i++) der[i] = t[2 * NB + i - 1];
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

__device__ __forceinline__ float rq_spline_inverse(float x, const float* h, float filter_sqrt) {
    if (!(x < 5.0f && x > -5.0f)) return x;
    float t[29];
#pragma unroll
    for (int j = 0; j < 29; j++) t[j] = rq_spline_param(j, h[j], filter_sqrt);
    return rq_spline_inverse_t(x, t);
}

// /root/reference/src/models/SynthesizerTrn.cpp:393-396: (int16_t)(o * 32737) -- no clip.  The reference build (x86-64,
// gcc) converts with cvttss2si to a 32-bit integer and keeps the low 16 bits: in-range values truncate toward zero,
// |v| < 2^31 wraps around modulo 2^16, anything beyond (and NaN) becomes 0x80000000 -> 0.  v_cvt_i32_f32 saturates
// instead, so the out-of-int32 case is mapped explicitly.
__device__ __forceinline__ int16_t pcm_cast(float o) {
    const float v = o * 32737.0f;
    const int32_t q = fabsf(v) < 2147483648.0f ? (int32_t)v : (int32_t)0x80000000;
    return (int16_t)q;
}

}  // namespace sts
