// Micro-benchmark: cost of a device-wide barrier inside one persistent kernel on MI355X (8 XCDs, private
// L2s), versus the back-to-back dispatch floor of dependent kernel launches (~4.4 us in the kernel trace).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned nblocks, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch += nblocks;
        __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);                       // agent scope release (L2 write-back)
        while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < epoch) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

// every iteration: each block writes a value, barrier, reads its neighbour's value (cross-XCD visibility check)
__global__ void persistent(unsigned* counter, float* buf, int iters, int* errors) {
    unsigned epoch = 0;
    const unsigned nb = gridDim.x;
    int bad = 0;
    for (int it = 0; it < iters; it++) {
        if (threadIdx.x < 64) buf[blockIdx.x * 64 + threadIdx.x] = (float)(it * 1000 + blockIdx.x);
        grid_barrier(counter, nb, epoch);
        const unsigned nbr = (blockIdx.x + 1) % nb;
        if (threadIdx.x < 64) {
            float v = __builtin_nontemporal_load(&buf[nbr * 64 + threadIdx.x]);
            if (v != (float)(it * 1000 + nbr)) bad++;
        }
        grid_barrier(counter, nb, epoch);
    }
    if (bad) atomicAdd(errors, bad);
}

__global__ void tiny(float* buf, int it) {
    if (threadIdx.x < 64) buf[blockIdx.x * 64 + threadIdx.x] = (float)it;
}

int main() {
    unsigned* counter; float* buf; int* errors;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&buf, 1024 * 64 * 4)); CK(hipMalloc(&errors, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 500;
    for (int threads : {256, 1024}) {
        for (int blocks : {64, 256, 512}) {
            CK(hipMemset(counter, 0, 4)); CK(hipMemset(errors, 0, 4));
            void* args[] = {&counter, &buf, (void*)&iters, &errors};
            CK(hipLaunchCooperativeKernel((void*)persistent, dim3(blocks), dim3(threads), args, 0, 0));   // warm-up
            CK(hipDeviceSynchronize());
            CK(hipMemset(counter, 0, 4));
            CK(hipEventRecord(e0));
            CK(hipLaunchCooperativeKernel((void*)persistent, dim3(blocks), dim3(threads), args, 0, 0));
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int err; CK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
            printf("persistent blocks=%d threads=%d: %.2f us per barrier (%d barriers), visibility errors=%d\n", blocks, threads,
                   1e3 * ms / (2 * iters), 2 * iters, err);
        }
    }
    CK(hipEventRecord(e0));
    for (int it = 0; it < 1000; it++) hipLaunchKernelGGL(tiny, dim3(256), dim3(256), 0, 0, buf, it);
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("back-to-back dependent launches: %.2f us per launch\n", 1e3 * ms / 1000);
    return 0;
}
