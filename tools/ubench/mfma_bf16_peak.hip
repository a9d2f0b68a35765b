// Micro-benchmark: what does v_mfma_f32_32x32x16_bf16 sustain on this MI355X with the split-bf16 conv kernel's instruction
// mix -- 4 accumulators, 6 products per operand set -- on constant operands, on random bf16 operands, and on operands with
// the statistics of a split fp32 number (hi / mid / lo planes of random fp32 data)?  Calibrates the roofline expectation of
// conv_bf3.hip: the data-dependent power draw of the matrix pipe lowers the clock (DVFS).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_bf16_peak.hip -o /tmp/mfma_bf16_peak && /tmp/mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned hash(unsigned h) { h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15; h *= 2654435761u; h ^= h >> 16; return h; }
// MODE 0: constant operands; 1: random bf16 bit patterns (exponent near 1.0); 2: the three planes of split random fp32 values
template <int MODE>
__device__ inline void make_planes(unsigned seed, u32x4 (&pl)[3]) {
    for (int d = 0; d < 4; d++) {
        unsigned w[3] = {0, 0, 0};
        for (int e = 0; e < 2; e++) {
            const unsigned h = hash(seed * 8u + d * 2 + e);
            unsigned short p[3];
            if (MODE == 0) { p[0] = 0x3f80; p[1] = 0x3c00; p[2] = 0x3800; }
            else if (MODE == 1) { p[0] = 0x3f00 | (h & 0xff); p[1] = 0x3b00 | ((h >> 8) & 0xff) | ((h >> 1) & 0x8000); p[2] = 0x3700 | ((h >> 16) & 0xff) | ((h >> 2) & 0x8000); }
            else {
                const float x = ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) * 1.7f;
                const unsigned u = __builtin_bit_cast(unsigned, x);
                const float r = x - __builtin_bit_cast(float, u & 0xffff0000u);
                const unsigned v = __builtin_bit_cast(unsigned, r);
                const float l = r - __builtin_bit_cast(float, v & 0xffff0000u);
                p[0] = u >> 16; p[1] = v >> 16; p[2] = __builtin_bit_cast(unsigned, l) >> 16;
            }
            for (int k = 0; k < 3; k++) w[k] |= (unsigned)p[k] << (16 * e);
        }
        for (int k = 0; k < 3; k++) pl[k][d] = w[k];
    }
}

template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float* out, int iters) {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int q = 0; q < 2; q++) for (int r = 0; r < 16; r++) acc[i][q][r] = 0.f;
    u32x4 a[2][3], b[2][3];
    for (int i = 0; i < 2; i++) { make_planes<MODE>(threadIdx.x * 4 + i + blockIdx.x * 1024, a[i]); make_planes<MODE>(threadIdx.x * 4 + 2 + i + blockIdx.x * 1024, b[i]); }
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int q = 0; q < 2; q++)
                    acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i][PA[p]]), __builtin_bit_cast(bf16x8, b[q][PB[p]]), acc[i][q], 0, 0, 0);
        if (MODE) {   // rotate the operands a little so that successive MFMAs do not see identical inputs
            for (int i = 0; i < 2; i++) for (int pl = 0; pl < 3; pl++) { a[i][pl] = a[i][pl].yzwx; b[i][pl] = b[i][pl].wxyz; }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; i++) for (int q = 0; q < 2; q++) for (int r = 0; r < 16; r++) s += acc[i][q][r];
    if (s == 123.456f) out[0] = s;
}

template <int MODE, int WAVES>
void run(const char* name, int blocks, int iters) {
    float* d; hipMalloc(&d, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, d, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * WAVES * iters * 24 * 2.0 * 32 * 32 * 16;
    printf("%-58s blocks=%5d  %8.3f ms  %7.1f bf16 TFLOP/s = %6.1f TFLOP/s fp32-equivalent (/6)\n", name, blocks, ms, flops / ms / 1e9, flops / ms / 1e9 / 6.0);
    hipFree(d);
}

// ---- the two-term fp16 form of conv_bf3.hip (MATH 1): 4 accumulators, 3 products per operand set = 12 v_mfma_f32_32x32x16_f16 per
// iteration, operands = fp16 planes of random fp32 values (weights: P0 / P1 / P2 = P0 2^-11 of w 2^13; activations: hi / lo' = (x - hi) 2^11)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int MODE>
__device__ inline void make_planes_h(unsigned seed, u32x4 (&pl)[3], bool weight) {
    for (int d = 0; d < 4; d++) {
        unsigned w[3] = {0, 0, 0};
        for (int e = 0; e < 2; e++) {
            const unsigned h = hash(seed * 8u + d * 2 + e);
            _Float16 p[3];
            if (MODE == 0) { p[0] = (_Float16)1.0f; p[1] = (_Float16)0.125f; p[2] = (_Float16)0.25f; }
            else {
                const float x = ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) * (weight ? 8000.0f : 1.7f);
                p[0] = (_Float16)x;
                if (weight) { p[1] = (_Float16)(x - (float)p[0]); p[2] = (_Float16)((float)p[0] * (1.0f / 2048.0f)); }
                else { p[1] = (_Float16)((x - (float)p[0]) * 2048.0f); p[2] = (_Float16)0.0f; }
            }
            for (int k = 0; k < 3; k++) w[k] |= (unsigned)__builtin_bit_cast(unsigned short, p[k]) << (16 * e);
        }
        for (int k = 0; k < 3; k++) pl[k][d] = w[k];
    }
}
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void kh(float* out, int iters) {
    f32x16 acc[2][2];
    for (int i = 0; i < 2; i++) for (int q = 0; q < 2; q++) for (int r = 0; r < 16; r++) acc[i][q][r] = 0.f;
    u32x4 a[2][3], b[2][3];
    for (int i = 0; i < 2; i++) { make_planes_h<MODE>(threadIdx.x * 4 + i + blockIdx.x * 1024, a[i], true); make_planes_h<MODE>(threadIdx.x * 4 + 2 + i + blockIdx.x * 1024, b[i], false); }
    constexpr int PA[3] = {2, 1, 0}, PB[3] = {1, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int i = 0; i < 2; i++)
#pragma unroll
                for (int q = 0; q < 2; q++)
                    acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[i][PA[p]]), __builtin_bit_cast(f16x8, b[q][PB[p]]), acc[i][q], 0, 0, 0);
        if (MODE) {
            for (int i = 0; i < 2; i++) for (int pl = 0; pl < 3; pl++) { a[i][pl] = a[i][pl].yzwx; b[i][pl] = b[i][pl].wxyz; }
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; i++) for (int q = 0; q < 2; q++) for (int r = 0; r < 16; r++) s += acc[i][q][r];
    if (s == 123.456f) out[0] = s;
}
// fp16 TFLOP/s of the 12-MFMA sequence (mode 0: constant operands, otherwise the two-term planes of random fp32 values)
extern "C" double sts_ubench_mfma_f16(int mode, int blocks, int iters) {
    float* d = nullptr;
    if (hipMalloc(&d, 4) != hipSuccess) return -1.0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&](int it) {
        if (mode == 0) hipLaunchKernelGGL((kh<0, 4>), dim3(blocks), dim3(256), 0, 0, d, it);
        else hipLaunchKernelGGL((kh<2, 4>), dim3(blocks), dim3(256), 0, 0, d, it);
    };
    launch(50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(iters);
    hipEventRecord(e1);
    float ms = 0.f;
    const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f;
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(d);
    if (!ok) return -1.0;
    return (double)blocks * 4 * iters * 12 * 2.0 * 32 * 32 * 16 / ms / 1e9;
}

// Library form (tools/ubench/Makefile -> libsts_ubench.so): bench.py calls this in its own process so that the bench line carries the
// ceiling measured on the box and at the moment of the run.  Returns bf16 TFLOP/s (<= 0 on failure).
extern "C" double sts_ubench_mfma_bf16(int mode, int blocks, int iters) {
    float* d = nullptr;
    if (hipMalloc(&d, 4) != hipSuccess) return -1.0;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto launch = [&](int it) {
        if (mode == 0) hipLaunchKernelGGL((k<0, 4>), dim3(blocks), dim3(256), 0, 0, d, it);
        else if (mode == 1) hipLaunchKernelGGL((k<1, 4>), dim3(blocks), dim3(256), 0, 0, d, it);
        else hipLaunchKernelGGL((k<2, 4>), dim3(blocks), dim3(256), 0, 0, d, it);
    };
    launch(50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(iters);
    hipEventRecord(e1);
    float ms = 0.f;
    const bool ok = hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms > 0.f;
    hipEventDestroy(e0); hipEventDestroy(e1); hipFree(d);
    if (!ok) return -1.0;
    return (double)blocks * 4 * iters * 24 * 2.0 * 32 * 32 * 16 / ms / 1e9;
}

#ifndef STS_UBENCH_LIB
int main() {
    run<0, 4>("constant operands, 4 waves/WG, 1 WG/CU", 256, 20000);
    run<0, 4>("constant operands, 4 waves/WG, 2 WG/CU", 512, 20000);
    run<1, 4>("random bf16 operands, 2 WG/CU", 512, 20000);
    run<2, 4>("planes of split random fp32 operands, 2 WG/CU", 512, 20000);
    run<2, 4>("planes of split random fp32 operands, 2 WG/CU, 10x longer", 512, 200000);
    run<2, 4>("planes of split random fp32 operands, 1 WG/CU", 256, 20000);
    return 0;
}
#endif
