// Micro-benchmark / coherence check for the persistent single-XCD kernels (round 3): after a barrier among workgroups of ONE XCD
// (no agent-scope release, see xcd_barrier.hip), which kinds of load see the other workgroups' stores?
//   mode 0: non-temporal loads (bypass the CU's vector L1)               -- what xcd_barrier.hip used
//   mode 1: plain loads after an agent-scope ACQUIRE fence (buffer_inv sc1: invalidates the wave's L1 view)
//   mode 2: plain loads, no invalidate                                     -- control: stale L1 lines must show up as errors
//   mode 3: plain loads with the sc1 bit (agent-scope load)
//   mode 4: read-modify-write across workgroups (row r is incremented by a different workgroup every round): NT load + store
//   mode 5: as 4 with plain loads after the acquire fence
// and what does each cost per barrier?  Also: a WORK-CLAIM barrier (tiles handed out through an atomic counter, the step ends when
// a completion counter reaches the tile count) -- the deadlock-free form: it never waits for a workgroup that is not resident.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/xcd_coherence.hip -o /tmp/xcd_coherence && /tmp/xcd_coherence
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ void xcd_barrier(unsigned* counter, unsigned nblocks, unsigned& epoch) {
    __syncthreads();
    if (threadIdx.x == 0) {
        epoch += nblocks;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}

template <int MODE>
__device__ __forceinline__ float ld(const float* p) {
    if (MODE == 0 || MODE == 4) return __builtin_nontemporal_load(p);
    if (MODE == 3) { float v; asm volatile("global_load_dword %0, %1, off sc1\n s_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
    return *(const volatile float*)p == 0.f ? *p : *p;      // plain load (volatile read first would itself be sc0 sc1: avoid) -- see below
}
template <> __device__ __forceinline__ float ld<1>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<2>(const float* p) { return *p; }
template <> __device__ __forceinline__ float ld<5>(const float* p) { return *p; }

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* counter, float* buf, int iters, int* errors, int xcd, int nwork) {
    if ((int)(blockIdx.x & 7) != xcd) return;
    const unsigned me = blockIdx.x >> 3;
    unsigned epoch = 0;
    int bad = 0;
    const int t = threadIdx.x;
    for (int it = 1; it <= iters; it++) {
        if (MODE < 4) {
            buf[me * 256 + t] = (float)(it * 64 + (int)me);
            xcd_barrier(counter, nwork, epoch);
            if (MODE == 1) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int kk = 0; kk < nwork; kk++) {
                const float v = ld<MODE>(&buf[kk * 256 + t]);
                if (v != (float)(it * 64 + kk)) bad++;
            }
            xcd_barrier(counter, nwork, epoch);
        } else {
            const unsigned row = (me + it) % nwork;             // every row gets exactly one increment per round, from a rotating owner
            if (MODE == 5) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            const float v = ld<MODE>(&buf[row * 256 + t]);
            buf[row * 256 + t] = v + 1.0f;
            xcd_barrier(counter, nwork, epoch);
        }
    }
    if (MODE >= 4) {
        if (MODE == 5) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const float v = ld<MODE == 5 ? 1 : 0>(&buf[me * 256 + t]);
        if (v != (float)iters) bad++;
    }
    if (bad) atomicAdd(errors, bad);
}

// work-claim step: `ntiles` tiles, a workgroup claims one at a time; the step is over when done == ntiles * (step + 1)
__global__ __launch_bounds__(256) void kclaim(unsigned* claim, unsigned* done, float* buf, int steps, int ntiles, int* errors, int xcd) {
    if ((int)(blockIdx.x & 7) != xcd) return;
    __shared__ unsigned s_tile;
    int bad = 0;
    const int t = threadIdx.x;
    for (int s = 0; s < steps; s++) {
        for (;;) {
            if (t == 0) s_tile = __hip_atomic_fetch_add(&claim[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const unsigned tile = s_tile;
            __syncthreads();
            if (tile >= (unsigned)ntiles) break;
            // the tile's "work": read the previous step's value of a DIFFERENT tile, write this step's value
            if (s > 0) { const float v = __builtin_nontemporal_load(&buf[((tile + 1) % ntiles) * 256 + t]); if (v < (float)(s - 1) * 64) bad++; }
            __syncthreads();
            if (s > 0) { const float v = __builtin_nontemporal_load(&buf[tile * 256 + t]); if (v != (float)((s - 1) * 64 + (int)tile)) bad++; }
            buf[tile * 256 + t] = (float)(s * 64 + (int)tile);
            __syncthreads();
            if (t == 0) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        if (t == 0) while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ntiles * (s + 1)) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
    }
    if (bad) atomicAdd(errors, bad);
}

template <int MODE>
void run(const char* name, int nwork, int iters) {
    unsigned* counter; float* buf; int* errors;
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&buf, 64 * 256 * 4)); CK(hipMalloc(&errors, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0; int err = 0;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipMemset(counter, 0, 4)); CK(hipMemset(errors, 0, 4)); CK(hipMemset(buf, 0, 64 * 256 * 4));
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k<MODE>, dim3(nwork * 8), dim3(256), 0, 0, counter, buf, iters, errors, 5, nwork);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    }
    CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
    printf("%-78s %2d WGs: %6.2f us per round, errors=%d\n", name, nwork, 1e3 * ms / iters, err);
    hipFree(counter); hipFree(buf); hipFree(errors);
}

int main() {
    const int iters = 3000;
    for (int nwork : {8, 32}) {
        run<0>("write / barrier / read ALL rows / barrier: non-temporal loads", nwork, iters);
        run<1>("write / barrier / read ALL rows / barrier: acquire fence (agent) + plain loads", nwork, iters);
        run<3>("write / barrier / read ALL rows / barrier: sc1 loads", nwork, iters);
        run<2>("write / barrier / read ALL rows / barrier: plain loads, NO invalidate (control)", nwork, iters);
        run<4>("rotating read-modify-write + one barrier: non-temporal load", nwork, iters);
        run<5>("rotating read-modify-write + one barrier: acquire fence (agent) + plain load", nwork, iters);
    }
    for (int ntiles : {24, 96, 400}) {
        unsigned *claim, *done; float* buf; int* errors; const int steps = 2000;
        CK(hipMalloc(&claim, steps * 4)); CK(hipMalloc(&done, 4)); CK(hipMalloc(&buf, 512 * 256 * 4)); CK(hipMalloc(&errors, 4));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms = 0; int err = 0;
        for (int rep = 0; rep < 2; rep++) {
            CK(hipMemset(claim, 0, steps * 4)); CK(hipMemset(done, 0, 4)); CK(hipMemset(errors, 0, 4));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(kclaim, dim3(32 * 8), dim3(256), 0, 0, claim, done, buf, steps, ntiles, errors, 5);
            CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        }
        CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipMemcpy(&err, errors, 4, hipMemcpyDeviceToHost));
        printf("work-claim step (atomic tile counter + completion counter), 32 WGs, %3d tiles per step: %6.2f us per step, errors=%d\n", ntiles, 1e3 * ms / steps, err);
    }
    return 0;
}
