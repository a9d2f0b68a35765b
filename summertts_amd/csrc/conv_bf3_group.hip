// conv_bf3_group.hip -- grouped launches of the split-operand convs: layer d of ALL ResBlock chains of a decoder stage in one grid
// (conv_bf3_group_kernel, conv_bf3_dev.hpp).  /root/reference/src/modules/ResBlock1.cpp:55-69, Generator_hifigan.cpp:151-175.
#include "conv_bf3_dev.hpp"

namespace sts {

#ifdef STS_TILE_TRACE
int tile_trace_bind_group(long long* buf, unsigned capacity_records) { return tile_trace_bind(buf, capacity_records); }
#endif

#ifdef STS_EXPERIMENTS   // lab build only: a tie with the grouped launches, not shipped
// ------------------------------------------------------------------------------------------------
// Persistent decoder-stage kernel (kernels.hpp StageArgs).  512 workgroups (two per CU) of the 128 x 128 tile; a workgroup serves
// the XCD it runs on.  Item t of an XCD = ((op * ncol + column) * nmem + chain): claimed in this order through an L2 atomic, so every
// item's dependencies -- the three column tiles around it of the previous op of the same chain -- were claimed earlier by
// workgroups that never wait for a later item: no deadlock, whatever else runs on the device.  Coherence as in persist.hip:
// in-kernel data is read with non-temporal loads (NTL body), a wave waits for its stores before the workgroup barrier that
// precedes the completion flag, flags are relaxed agent-scope atomics in the XCD's L2.
// ------------------------------------------------------------------------------------------------
constexpr int PS_MAX_CONVS = 32;      // convs (ops x chains) of one stage whose descriptors are kept in LDS
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void conv_bf3_stage_kernel(StageArgs A) {
    __shared__ int s_item;
    __shared__ ConvArgs s_tab[PS_MAX_CONVS];        // this XCD's conv descriptors: an item reads its own out of LDS
    const int tid = threadIdx.x;
    const int x = (int)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 7u);
    const int ncol = A.ncol[x], nmem = A.nmem, nops = A.nops;
    const int total = nops * ncol * nmem;
    const size_t per_xcd = (size_t)1 + (size_t)nops * nmem * (PS_MAX_COLS + 1);
    unsigned* claim = A.ctr + (size_t)x * per_xcd;
    unsigned* done = claim + 1;                     // [op][chain][PS_MAX_COLS flags | 1 count]
    unsigned next = 0;
    if (tid == 0) next = __hip_atomic_fetch_add(claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef STS_TILE_TRACE
    long long ps_t0 = (long long)__builtin_amdgcn_s_memtime(), ps_wait = 0, ps_items = 0;
#endif
    {
        const int nw = nops * nmem * (int)(sizeof(ConvArgs) / 4);
        const int* src = (const int*)(A.tab + (size_t)x * nops * nmem);
        for (int i = tid; i < nw; i += 256) ((int*)s_tab)[i] = src[i];
    }
    for (;;) {
        if (tid == 0) s_item = (int)next;
        __syncthreads();
        const int t = __builtin_amdgcn_readfirstlane(s_item);
        if (t >= total) break;
        if (tid == 0) next = __hip_atomic_fetch_add(claim, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // the next item: its round trip rides under this tile
        const int op = t / (ncol * nmem), rem = t - op * ncol * nmem;
        const int col = rem / nmem, m = rem - col * nmem;
#ifdef STS_TILE_TRACE
        const long long ps_w0 = (long long)__builtin_amdgcn_s_memtime();
#endif
        if (op > 0 && tid == 0) {
            // usually the whole previous op of this chain is complete (one count to look at); otherwise the three column tiles
            // this one reads, all three flags requested together
            const unsigned* f = done + ((size_t)(op - 1) * nmem + m) * (PS_MAX_COLS + 1);
            if (__hip_atomic_load(f + PS_MAX_COLS, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)ncol) {
                const int c0 = col > 0 ? col - 1 : col, c1 = col + 1 < ncol ? col + 1 : col;
                for (;;) {
                    const unsigned a0 = __hip_atomic_load(f + c0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned a1 = __hip_atomic_load(f + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const unsigned a2 = __hip_atomic_load(f + c1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (a0 && a1 && a2) break;
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        __syncthreads();
#ifdef STS_TILE_TRACE
        ps_wait += (long long)__builtin_amdgcn_s_memtime() - ps_w0; ps_items++;
#endif
        // the conv's descriptor word by word through readfirstlane: every field in scalar registers (read through a pointer the
        // compiler treats the fields as divergent and wraps each buffer load in a waterfall loop: 22 per K step, measured 1.9x slower)
        ConvArgs al;
        {
            const int* src = (const int*)&s_tab[op * nmem + m];
            int* dstw = (int*)&al;
            static_assert(sizeof(ConvArgs) % 4 == 0, "ConvArgs is copied in 32-bit words");
#pragma unroll
            for (int w = 0; w < (int)(sizeof(ConvArgs) / 4); w++) dstw[w] = __builtin_amdgcn_readfirstlane(src[w]);
        }
#ifndef STS_STAGE_NTL
#define STS_STAGE_NTL true
#endif
        conv_bf3_body<2, 2, 2, 2, 1, 1, STS_STAGE_NTL>(al, 1, col, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's stores have reached the L2
        __syncthreads();                                           // ... and everybody's; the staged LDS window is free again
        if (tid == 0) {
            unsigned* f = done + ((size_t)op * nmem + m) * (PS_MAX_COLS + 1);
            __hip_atomic_store(f + col, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(f + PS_MAX_COLS, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#ifdef STS_TILE_TRACE
    if (long long* r = tt_open(2, x)) {      // kind 2 = one persistent stage workgroup: lifetime, dependency wait, items
        r[4] = ps_t0; r[5] = (long long)__builtin_amdgcn_s_memtime(); r[6] = ps_wait; r[7] = ps_items; r[8] = 0; r[9] = 0;
        r[10] = (long long)__builtin_amdgcn_s_memrealtime();
    }
#endif
    // the last workgroup to leave re-arms the counters for the next launch
    __syncthreads();
    unsigned* exitc = A.ctr + 8 * per_xcd;
    if (tid == 0) s_item = __hip_atomic_fetch_add(exitc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (s_item)
        for (size_t i = tid; i <= 8 * per_xcd; i += 256) __hip_atomic_store(A.ctr + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

#endif  // STS_EXPERIMENTS

template <int MW, int NW, int WM, int WN, int NSUB, int KG = 1, bool H2 = false>
static void launch_bf3_group(const ConvGroup& G, hipStream_t st) {
    constexpr int MT = 32 * MW * WM, NT = 32 * NW * WN;
    const ConvArgs& a = G.g[0];
    const int mt = (a.Cout_pad + MT - 1) / MT;
    const int nx = (a.max_n + NT - 1) / NT;
    static const int il = exp_int("STS_BF3_INTERLEAVE", 0);
    if constexpr (H2) {
        if (a.math == 1) {
            const size_t lds = bf3_lds_bytes<MW, NW, WM, WN, NSUB, KG, 1>();
            hipLaunchKernelGGL((conv_bf3_group_kernel<MW, NW, WM, WN, NSUB, KG, 1>), dim3(mapped_grid(nx, mt, a.B * G.n)), dim3(WM * WN * KG * 64), lds, st, G,
                               mt, a.B, nx, mt, il & 1);
            return;
        }
    }
    const size_t lds = bf3_lds_bytes<MW, NW, WM, WN, NSUB, KG>();
    hipLaunchKernelGGL((conv_bf3_group_kernel<MW, NW, WM, WN, NSUB, KG>), dim3(mapped_grid(nx, mt, a.B * G.n)), dim3(WM * WN * KG * 64), lds, st, G,
                       mt, a.B, nx, mt, il & 1);
}

#ifdef STS_EXPERIMENTS
size_t ps_counter_bytes(int nops, int nmem) { return ((size_t)8 * (1 + (size_t)nops * nmem * (PS_MAX_COLS + 1)) + 16) * sizeof(unsigned); }
// the stage kernel is instantiated for the plain 128 x 128 tile: one 128-row tile, 16-channel staged chunks, a plain (non-polyphase) conv
bool conv_bf3_stage_eligible(const ConvArgs& a) {
    return conv_bf3_eligible(a) && !a.transposed && a.Cout_pad == 128 && a.out_stride == 1 && (a.epi == EPI_STORE || a.epi == EPI_RESADD) && !a.ubias;
}
void conv_bf3_stage(const StageArgs& A, hipStream_t st) {
    if (A.nops * A.nmem > PS_MAX_CONVS) return;          // (the engine checks the same bound before it builds the table)
    const size_t lds = bf3_lds_bytes<2, 2, 2, 2, 1, 1>();
    hipLaunchKernelGGL(conv_bf3_stage_kernel, dim3(512), dim3(256), lds, st, A);
}
#endif  // STS_EXPERIMENTS

bool conv_bf3_group_eligible(const ConvGroup& G) {
    if (G.n < 1 || G.n > kMaxGroup) return false;
    const ConvArgs& r = G.g[0];
    for (int i = 0; i < G.n; i++) {
        const ConvArgs& a = G.g[i];
        if (!conv_bf3_eligible(a) || a.transposed || a.epi == EPI_GATE) return false;
        if (a.Cout_pad != r.Cout_pad || a.max_n != r.max_n || a.B != r.B || a.math != r.math) return false;
    }
    return true;
}

void conv_bf3_group(const ConvGroup& Gin, hipStream_t st, int tile) {
    ConvGroup G = Gin;
    if (G.g[0].max_n <= 0 || G.g[0].B <= 0) return;
    for (int i = 1; i < G.n; i++)                       // longest K loop first
        for (int j = i; j > 0 && (long)G.g[j].ntap * G.g[j].Cin_pad > (long)G.g[j - 1].ntap * G.g[j - 1].Cin_pad; j--) {
            ConvArgs t = G.g[j]; G.g[j] = G.g[j - 1]; G.g[j - 1] = t;
        }
    const ConvArgs& a = G.g[0];
    if (!bf3_tile_ok(tile) || (tile >= 21 && tile != 24) || (a.math == 1 && !h2_tile(tile))) tile = pick_bf3_tile(a.Cout_pad, a.max_n, (long)a.B * G.n, false, a.math);
    if (tile >= 8 && tile < 16) for (int i = 0; i < G.n; i++) if (G.g[i].Cin_pad % 32 != 0) { tile -= 8; break; }
    switch (tile) {
        case 20: { bool ok = true; for (int i = 0; i < G.n; i++) ok = ok && G.g[i].Cin_pad % 32 == 0;
                   if (ok) launch_bf3_group<2, 2, 2, 2, 2, 2, true>(G, st); else launch_bf3_group<2, 2, 2, 2, 1, 1, true>(G, st); break; }
#ifdef STS_EXPERIMENTS
        case 24: { bool ok = true; for (int i = 0; i < G.n; i++) ok = ok && G.g[i].Cin_pad % 32 == 0;
                   if (ok) launch_bf3_group<2, 2, 4, 1, 2, 2, true>(G, st); else launch_bf3_group<2, 2, 2, 2, 1, 1, true>(G, st); break; }
#endif
        case 0: launch_bf3_group<2, 2, 2, 2, 1, 1, true>(G, st); break;
        case 3: launch_bf3_group<2, 2, 1, 2, 1, 1, true>(G, st); break;
#ifdef STS_EXPERIMENTS      // tiles no automatic choice selects (measured ties / losses, profiles/r03_*tile_sweep.log): lab build only
        case 1: launch_bf3_group<2, 2, 1, 4, 1>(G, st); break;
        case 2: launch_bf3_group<2, 2, 2, 4, 1>(G, st); break;
        case 5: launch_bf3_group<1, 2, 1, 2, 1>(G, st); break;
        case 6: launch_bf3_group<1, 4, 4, 1, 1>(G, st); break;
        case 7: launch_bf3_group<1, 4, 2, 1, 1>(G, st); break;
        case 14: launch_bf3_group<1, 4, 4, 1, 2>(G, st); break;
        case 15: launch_bf3_group<1, 4, 2, 1, 2>(G, st); break;
        case 8: launch_bf3_group<2, 2, 2, 2, 2>(G, st); break;
        case 9: launch_bf3_group<2, 2, 1, 4, 2>(G, st); break;
        case 10: launch_bf3_group<2, 2, 2, 4, 2>(G, st); break;
        case 11: launch_bf3_group<2, 2, 1, 2, 2>(G, st); break;
        case 12: launch_bf3_group<1, 2, 1, 4, 2>(G, st); break;
        case 13: launch_bf3_group<1, 2, 1, 2, 2>(G, st); break;
#endif
        default: launch_bf3_group<1, 2, 1, 4, 1, 1, true>(G, st); break;       // 4: 32 x 256
    }
}

}  // namespace sts
