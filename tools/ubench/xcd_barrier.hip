// Micro-benchmark: a barrier among workgroups that all sit on ONE XCD (workgroup ids congruent modulo 8 -- the dispatch deals
// ids round-robin over the 8 XCDs, which conv_common.hpp's tile map already relies on), through an atomic counter in that
// XCD's L2, WITHOUT an agent-scope release (no L2 write-back: the XCD's own L2 is the point of coherence for its 32 CUs).
// Question for the next round: could the launch-bound front (text encoder + duration predictor: ~75 dependent launches of
// ~8 us on 8..250 workgroups each) run as one persistent single-XCD kernel with such barriers between its layers?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/xcd_barrier.hip -o /tmp/xcd_barrier && /tmp/xcd_barrier
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xf; }   // HW_REG_XCC_ID[3:0]

__device__ __forceinline__ void xcd_barrier(unsigned* counter, unsigned nblocks, unsigned& epoch) {
    __syncthreads();                                            // the block's stores are issued ...
    if (threadIdx.x == 0) {
        epoch += nblocks;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // ... and complete (vmcnt(0)); no L2 write-back
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

__global__ void persistent(unsigned* counter, float* buf, int iters, int* errors, unsigned* xcc, int xcd, int nwork) {
    if ((int)(blockIdx.x & 7) != xcd) return;
    const unsigned me = blockIdx.x >> 3;                        // 0 .. nwork-1
    if (threadIdx.x == 0) xcc[me] = xcc_id();
    unsigned epoch = 0;
    int bad = 0;
    for (int it = 0; it < iters; it++) {
        if (threadIdx.x < 64) buf[me * 64 + threadIdx.x] = (float)(it * 1000 + me);
        xcd_barrier(counter, nwork, epoch);
        const unsigned nbr = (me + 1) % nwork;
        if (threadIdx.x < 64) {
            const float v = __builtin_nontemporal_load(&buf[nbr * 64 + threadIdx.x]);     // bypasses this CU's L1
            if (v != (float)(it * 1000 + nbr)) bad++;
        }
        xcd_barrier(counter, nwork, epoch);
    }
    if (bad) atomicAdd(errors, bad);
}

int main() {
    unsigned *counter, *xcc; float* buf; int* errors;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&buf, 1024 * 64 * 4)); CK(hipMalloc(&errors, 4)); CK(hipMalloc(&xcc, 1024 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int threads : {256, 768}) {
        for (int nwork : {8, 32, 64}) {
            const int xcd = 3, blocks = nwork * 8;
            for (int rep = 0; rep < 2; rep++) {
                CK(hipMemset(counter, 0, 4)); CK(hipMemset(errors, 0, 4));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(persistent, dim3(blocks), dim3(threads), 0, 0, counter, buf, iters, errors, xcc, xcd, nwork);
                CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
            }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int err; CK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
            unsigned ids[64]; CK(hipMemcpy(ids, xcc, nwork * 4, hipMemcpyDeviceToHost));
            int same = 1; for (int i = 1; i < nwork; i++) same &= ids[i] == ids[0];
            printf("single-XCD barrier: %2d workgroups x %4d threads: %.2f us per barrier (%d barriers), visibility errors=%d, all on XCC %u: %s\n",
                   nwork, threads, 1e3 * ms / (2 * iters), 2 * iters, err, ids[0], same ? "yes" : "NO");
        }
    }
    return 0;
}
