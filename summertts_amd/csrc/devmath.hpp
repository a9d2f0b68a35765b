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
