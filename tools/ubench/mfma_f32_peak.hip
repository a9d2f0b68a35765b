// Micro-benchmark: what does v_mfma_f32_32x32x2_f32 sustain on this MI355X with the conv kernel's
// accumulator count, with and without the LDS fragment reads?  (calibrates the roofline expectation)
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_f32_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int LDS, int BAR, int RND>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ float sm[16 * 320];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16 * 320; i += 256) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;   // random-looking operands: constant data
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;                    // lets the chip clock ~15-20 % higher
        sm[i] = RND ? ((float)(h & 0xFFFFFF) / 8388608.0f - 1.0f) : (float)(i & 7) * 0.125f;
    }
    __syncthreads();
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) acc[a][r] = 0.f;
    float av = RND ? sm[(lane * 37) % 5000] : 0.5f + lane * 0.001f, bv[NACC];
    for (int a = 0; a < NACC; a++) bv[a] = 0.25f + a;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int p = 0; p < 8; p++) {
            if (LDS) {
#pragma unroll
                for (int a = 0; a < NACC; a++) bv[a] = sm[(2 * p + (lane >> 5)) * 320 + (lane & 31) + a * 32 + (it & 15)];
            }
            if (RND) av = sm[(lane * 37 + p * 11 + it) % 5000];
#pragma unroll
            for (int a = 0; a < NACC; a++) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[a], acc[a], 0, 0, 0);
        }
        if (BAR && (it % 11) == 10) __syncthreads();
    }
    float s = 0.f;
    for (int a = 0; a < NACC; a++) for (int r = 0; r < 16; r++) s += acc[a][r];
    if (s == 123.456f) out[0] = s;
}

template <int NACC, int LDS, int BAR, int RND = 0>
void run(const char* name, int blocks) {
    float* d; hipMalloc(&d, 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, LDS, BAR, RND>), dim3(blocks), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, LDS, BAR, RND>), dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * iters * 8 * NACC * 2.0 * 32 * 32 * 2;
    printf("%-34s blocks=%5d  %8.3f ms  %7.1f TFLOP/s\n", name, blocks, ms, flops / ms / 1e9);
    hipFree(d);
}

int main() {
    run<2, 0, 0>("2 acc, regs only", 256);
    run<2, 0, 0>("2 acc, regs only", 512);
    run<2, 0, 0>("2 acc, regs only", 1024);
    run<1, 0, 0>("1 acc, regs only", 512);
    run<4, 0, 0>("4 acc, regs only", 512);
    run<2, 1, 0>("2 acc + LDS B reads", 512);
    run<2, 1, 1>("2 acc + LDS B reads + barrier/11", 512);
    run<2, 1, 1>("2 acc + LDS + barrier, 668 blocks", 668);
    run<4, 1, 1>("4 acc + LDS + barrier", 512);
    run<2, 1, 1, 1>("2 acc + LDS + barrier, RANDOM data", 512);
    run<2, 1, 1, 1>("2 acc + LDS + barrier, RANDOM data", 2048);
    run<4, 1, 1, 1>("4 acc + LDS + barrier, RANDOM data", 2048);
    return 0;
}
