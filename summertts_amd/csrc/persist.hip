// persist.hip -- the reverse normalising flow of ONE utterance as ONE persistent launch (round 3).
//
// Reference: /root/reference/src/modules/ResidualCouplingBlock.cpp:59-70 (reverse order, flips),
// ResidualCouplingLayer.cpp:47-66 (x1 -= post(WN(pre(x0)))), WN.cpp:100-149 (gated layers, res/skip split) and the length
// regulator in front of it (SynthesizerTrn.cpp:304-321).
//
// Why: at one utterance the flow is 40 dependent convs of 9-14 us each on a few hundred frames -- every one of them a
// chip-wide launch whose price is the launch itself.  A workgroup-level barrier across the chip costs more than a
// launch (L2 write-back between the 8 XCDs), but a barrier among the workgroups of ONE XCD costs 0.5-0.7 us: their common L2 is
// the point of coherence (tools/ubench/xcd_barrier.hip, xcd_coherence.hip).  And the flow does not need the XCDs to talk:
// every op is a convolution along time with a small radius, so the frame axis is cut into 8 slices, one per XCD, each extended by
// the receptive field the REMAINING ops still need (2 frames per WaveNet layer: 32 frames for 4 couplings x 4 layers, shrinking
// to 0 at the last op).  Each XCD runs the whole flow on its own window in private buffers that never leave its L2, and only the
// final z columns it owns are written to the shared output.  Halo columns are computed twice (by both neighbours) from the same
// inputs in the same order, so the result does not depend on the cut.
//
// Execution model: 256 workgroups x 16 waves, one per CU.  A workgroup reads which XCD it runs on (HW_REG_XCC_ID) and serves
// that XCD's window.  Per op the window's output tiles (32 rows x 32 frames) are grouped into <= 32 chunks; a workgroup CLAIMS a
// chunk through an atomic counter in L2 (the claim for op s + 1 is issued while op s is being computed), processes it -- 16 / ks
// tiles at a time, ks waves splitting K of a tile, partial sums combined through LDS, the conv's epilogue fused --, bumps a
// completion counter, and waits until the op's chunks are all complete.  Nothing ever waits for a workgroup that is not
// resident: if chunks stay unclaimed (another kernel holds CUs) the waiting workgroups claim them after a few microseconds,
// so the scheme cannot deadlock whatever else runs on the device (sts_pool runs several engines on one GPU).
//
// Coherence rules inside the kernel (measured in xcd_coherence.hip: plain loads DO return stale L1 lines):
//   * every load of data written inside this kernel is NON-TEMPORAL (misses the CU's vector L1, served by the XCD's L2);
//     weights / biases / m / cum (constant for the kernel's lifetime) use ordinary cached loads;
//   * a wave waits for its stores (s_waitcnt vmcnt(0)) before the workgroup barrier that precedes the completion signal;
//   * counters are relaxed agent-scope atomics (executed in L2); no agent-scope fence anywhere (an acquire fence costs
//     more than the barrier itself: 10.9 vs 4.9 us per round in the micro-benchmark).
// Lab build only (-DSTS_EXPERIMENTS, `make exp`): measured slower than the launch-per-layer path (0.50 vs 0.44 ms), so the
// shipped library does not carry it -- no pk_* symbol, no extra weight copy, no sts_debug_set key.
#ifdef STS_EXPERIMENTS
#include "kernels.hpp"
#include "devmath.hpp"
#include "conv_common.hpp"

#include <string.h>

namespace sts {

constexpr int PK_WAVES = 16;
constexpr int PK_CHUNKS = 32;          // chunks per op and XCD (= workgroups of an XCD)
constexpr int PK_G = 4;                // K walked in groups of 4 channel pairs (8 input channels of one tap)
constexpr int PK_D = 4;                // register ring depth (groups in flight per wave)

__device__ __forceinline__ unsigned pk_xcc_id() { return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xf; }   // HW_REG_XCC_ID[3:0]
__device__ __forceinline__ float pk_ld(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void pk_stores_done() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

struct PkWin {      // this XCD's window, in frames of the utterance
    int f0, f1;     // frames it owns
    int fa, fb;     // frames its private buffers hold: [fa, fb) = own +- the total halo, clipped to the utterance
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Private activations are QUAD-INTERLEAVED: element (channel c, frame j) of a buffer lives at ((c >> 2) * wld + j) * 4 + (c & 3).
// A lane of an MFMA B operand needs 4 consecutive channels of one frame (one 16-byte load, 512 contiguous bytes per half-wave),
// and a lane of the 32x32 accumulator holds 4 consecutive rows of one frame (one 16-byte store).
__device__ __forceinline__ size_t pk_idx(int c, int j, int wld) { return ((size_t)(c >> 2) * wld + j) * 4 + (c & 3); }

// ---- op: length regulator into the private z window: zbuf(c, j) = m[c][phoneme(fa + j)]  (0 past the last duration)
__device__ __forceinline__ void pk_expand(const PkFlowArgs& A, const PkWin& W, float* zbuf, int chunk, int tid) {
    const int wlen = W.fb - W.fa;
    const int nq = (A.C + 3) / 4, q_per = (nq + PK_CHUNKS - 1) / PK_CHUNKS;
    const int q0 = chunk * q_per, q1 = q0 + q_per < nq ? q0 + q_per : nq;
    for (int j = tid; j < wlen; j += PK_WAVES * 64) {
        const int f = W.fa + j;
        int lo = 0, hi = A.T;       // first i with cum[i] > f
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (A.cum[mid] > f) hi = mid; else lo = mid + 1; }
        for (int q = q0; q < q1; q++) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = (lo < A.T && 4 * q + e < A.C) ? A.m[(size_t)(4 * q + e) * A.m_ld + lo] : 0.f;
            *(f32x4*)(zbuf + ((size_t)q * A.wld + j) * 4) = v;
        }
    }
}

// ---- op: the frames this XCD owns go to the shared z (plain [C][z_ld] layout)
__device__ __forceinline__ void pk_store_out(const PkFlowArgs& A, const PkWin& W, const float* zbuf, int chunk, int tid) {
    const int nq = (A.C + 3) / 4, q_per = (nq + PK_CHUNKS - 1) / PK_CHUNKS;
    const int q0 = chunk * q_per, q1 = q0 + q_per < nq ? q0 + q_per : nq;
    const int n = W.f1 - W.f0;
    for (int i = tid; i < n * (q1 - q0); i += PK_WAVES * 64) {
        const int q = q0 + i / n, j = i - (i / n) * n;
        const f32x4 v = __builtin_nontemporal_load((const f32x4*)(zbuf + ((size_t)q * A.wld + (W.f0 - W.fa) + j) * 4));
#pragma unroll
        for (int e = 0; e < 4; e++) if (4 * q + e < A.C) A.z[(size_t)(4 * q + e) * A.z_ld + W.f0 + j] = v[e];
    }
}

// ---- op: one chunk of a conv.  Tiles [chunk * per, ...) of the op's nt = mt * ncol tiles (column index fastest, so the tiles a
// workgroup holds at one time mostly share their weight rows); wave group g = wave / ks owns one tile per round, its ks waves
// take the K groups kw, kw + ks, ...  A group = 8 input channels of one tap = four v_mfma_f32_32x32x2_f32; lanes of the lower
// half-wave carry channels 0..3 of the group, the upper half channels 4..7 (MFMA i multiplies channel i of either half), so each
// operand of a group is ONE 16-byte load per lane: weights from the [tap][Cin/8][Cout][8] copy (1 KB contiguous per wave), inputs
// from the quad-interleaved private buffer through non-temporal loads.  Register ring of PK_D groups, no LDS / barrier in the loop.
struct PkGeo { int n_lo, n_hi, ncol, nt, per, ks, nch, pad; };      // geometry of one op on this XCD's window (computed once, kept in LDS)
__device__ __forceinline__ int pk_u(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <typename T> __device__ __forceinline__ T* pk_up(T* p) {
    const unsigned long long v = (unsigned long long)p;
    return (T*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}
__device__ __forceinline__ void pk_conv_chunk(const PkFlowArgs& A, const PkStep& sl, const PkGeo& gl, const PkWin& W, float* priv, const int chunk,
                                              float* red, const int tid, long long* tr) {
    if (tr && tid == 0) tr[4] = (long long)__builtin_amdgcn_s_memtime();
    // uniform copies of the op (LDS -> scalar registers): descriptors and scalar offsets below need them in SGPRs
    PkStep st;
    st.kind = PK_CONV; st.halo = 0;
    st.w = pk_up(sl.w); st.bias = pk_up(sl.bias); st.ubias_off = pk_u(sl.ubias_off);
    st.Cin = pk_u(sl.Cin); st.Cout = pk_u(sl.Cout); st.Cin_pad = pk_u(sl.Cin_pad); st.Cout_pad = pk_u(sl.Cout_pad);
    st.ntap = pk_u(sl.ntap); st.tap_step = pk_u(sl.tap_step); st.tap_off = pk_u(sl.tap_off);
    st.epi = pk_u(sl.epi); st.epi_flag = pk_u(sl.epi_flag); st.H = pk_u(sl.H); st.gate_perm = 0;
    st.in_buf = pk_u(sl.in_buf); st.in_row = pk_u(sl.in_row); st.out_buf = pk_u(sl.out_buf); st.out_row = pk_u(sl.out_row); st.aux_buf = pk_u(sl.aux_buf);
    const int n_lo = pk_u(gl.n_lo), n_hi = pk_u(gl.n_hi), ncol = pk_u(gl.ncol), nt = pk_u(gl.nt), per = pk_u(gl.per), ks = pk_u(gl.ks);
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int lks = 31 - __builtin_clz((unsigned)ks);     // ks is a power of two
    const int G = PK_WAVES >> lks;
    const int g = wave >> lks, kw = __builtin_amdgcn_readfirstlane(wave & (ks - 1));
    const int wlen = W.fb - W.fa;
    const size_t bstride = (size_t)A.rows * A.wld;
    const float* xin = priv + (size_t)st.in_buf * bstride + (size_t)st.in_row * A.wld;       // in_row is a multiple of 4
    float* yout = priv + (size_t)st.out_buf * bstride + (size_t)st.out_row * A.wld;
    float* aux = priv + (size_t)st.aux_buf * bstride;
    const int cinq = (st.Cin + 3) >> 2;                                                     // channel quads of the input
    const rsrc_t xrs = make_rsrc(xin, (unsigned)((size_t)cinq * A.wld * 16));
    const rsrc_t wrs = make_rsrc(st.w, (unsigned)((size_t)st.ntap * st.Cin_pad * st.Cout_pad * 4));
    const unsigned q16 = (unsigned)A.wld * 16u;                                              // bytes per channel quad
    const int gpt = st.Cin_pad / 8, gall = st.ntap * gpt;
    const int jstep = ks / gpt, cstep = ks - jstep * gpt;                                    // in groups
    const float* ub = st.ubias_off >= 0 ? A.cond + st.ubias_off : nullptr;
    const bool gate = st.epi == EPI_GATE;
    const bool split = st.epi == EPI_RESSKIP && st.Cout != st.H;      // rows < H update the input of the next layer, rows >= H the skip sum

    for (int r0 = 0; r0 < per; r0 += G) {
        const int t = chunk * per + r0 + g;
        const bool tv = t < nt && r0 + g < per;
        const int trow = tv ? t / ncol : 0, tcol = tv ? t - trow * ncol : 0;
        const int m0 = trow * 32, n0 = n_lo + tcol * 32;
        const int n = n0 + l31;
        // ---- epilogue operands.  This wave finishes the quads q = kw, kw + ks, ... (< 4; gate: < 2) of its group's tile; quad q =
        // accumulator registers 4q..4q+3 = rows m0 + 8q + 4 half + (0..3).  They are handled two at a time; the first pair's bias /
        // old values are requested BEFORE the K loop (ks >= 2: that is all there is), so they have landed when the sums are ready
        const int nq = gate ? 2 : 4;
        const bool ew = tv && kw < nq && n < n_hi;
        f32x4 bv[2], old[2];      // old: previous value of a read-modify-write epilogue | the sigmoid half's bias of a gate
        float* dst[2];
        auto load_ops = [&](int j0) {
#pragma unroll
            for (int qi = 0; qi < 2; qi++) {
                const int q = kw + (j0 + qi) * ks;
                bv[qi] = old[qi] = f32x4{0.f, 0.f, 0.f, 0.f};
                dst[qi] = nullptr;
                if (!ew || q >= nq) continue;
                const int row0 = m0 + 8 * q + 4 * half;
                if (gate) {
                    if (st.bias) { bv[qi] = *(const f32x4*)(st.bias + row0); old[qi] = *(const f32x4*)(st.bias + row0 + 16); }
                    if (ub) { bv[qi] += *(const f32x4*)(ub + row0); old[qi] += *(const f32x4*)(ub + row0 + 16); }
                    const int ch0 = (row0 >> 5) * 16 + (row0 & 15);
                    if (ch0 < st.H) dst[qi] = yout + pk_idx(ch0, n, A.wld);
                } else if (row0 < st.Cout) {
                    if (st.bias) bv[qi] = *(const f32x4*)(st.bias + row0);
                    if (ub) bv[qi] += *(const f32x4*)(ub + row0);
                    bool rd;
                    if (st.epi == EPI_RESSKIP && !(split && row0 < st.H)) { dst[qi] = aux + pk_idx(split ? row0 - st.H : row0, n, A.wld); rd = !(st.epi_flag & 1); }
                    else { dst[qi] = yout + pk_idx(row0, n, A.wld); rd = st.epi != EPI_STORE; }
                    if (rd) old[qi] = __builtin_nontemporal_load((const f32x4*)dst[qi]);
                }
            }
        };
        load_ops(0);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; r++) acc[r] = 0.f;
        if (tv) {
            const unsigned a_lane = (unsigned)(((size_t)(m0 + l31) * 8 + 4 * half) * 4);
            f32x4 ra[PK_D], rb[PK_D];
            int lj = kw / gpt, lg = kw - lj * gpt;          // (tap, group inside the tap) of the next group to load
            auto load_group = [&](f32x4& fa, f32x4& fb) {
                const int pos = n + lj * st.tap_step + st.tap_off;
                const bool v = pos >= 0 && pos < wlen && 8 * lg + 4 * half < st.Cin;
                const unsigned off = v ? (unsigned)pos * 16u + (unsigned)half * q16 : kOOB;
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                fb = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)off, (int)((unsigned)(2 * lg) * q16), 2));      // nt
                fa = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)a_lane, (int)((unsigned)((lj * gpt + lg) * st.Cout_pad) * 32u), 0));
                lj += jstep; lg += cstep;
                if (lg >= gpt) { lg -= gpt; lj++; }
            };
            int gnext = kw;
#pragma unroll
            for (int d = 0; d < PK_D; d++) { if (gnext < gall) load_group(ra[d], rb[d]); gnext += ks; }
            int gcur = kw;
            while (gcur < gall) {
#pragma unroll
                for (int d = 0; d < PK_D; d++) {
                    if (gcur < gall) {
#pragma unroll
                        for (int i = 0; i < 4; i++) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[d][i], rb[d][i], acc, 0, 0, 0);
                        if (gnext < gall) load_group(ra[d], rb[d]);
                    }
                    gnext += ks; gcur += ks;
                }
            }
        }
        if (tr && tid == 0 && r0 == 0) tr[5] = (long long)__builtin_amdgcn_s_memtime();
        // ---- combine the ks partial tiles of every group through LDS: red[wave][quad][lane] as 16-byte vectors
        static_for<0, 4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            f32x4 v = {acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]};
            *(f32x4*)(red + (((size_t)wave * 4 + q) * 64 + lane) * 4) = v;
        });
        __syncthreads();
        if (tr && tid == 0 && r0 == 0) tr[6] = (long long)__builtin_amdgcn_s_memtime();
        if (ew) {
            for (int j0 = 0; kw + j0 * ks < nq; j0 += 2) {
                if (j0) load_ops(j0);
#pragma unroll
                for (int qi = 0; qi < 2; qi++) {
                    const int q = kw + (j0 + qi) * ks;
                    if (q >= nq || !dst[qi]) continue;
                    f32x4 v = {0.f, 0.f, 0.f, 0.f}, v2 = {0.f, 0.f, 0.f, 0.f};
                    for (int k = 0; k < ks; k++) {
                        v += *(const f32x4*)(red + (((size_t)(g * ks + k) * 4 + q) * 64 + lane) * 4);
                        if (gate) v2 += *(const f32x4*)(red + (((size_t)(g * ks + k) * 4 + q + 2) * 64 + lane) * 4);
                    }
                    v += bv[qi];
                    if (gate) {
                        v2 += old[qi];
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; e++) o[e] = tanh_ref(v[e]) * sigmoid_ref(v2[e]);
                        *(f32x4*)dst[qi] = o;
                    } else if (st.epi == EPI_SUB) *(f32x4*)dst[qi] = old[qi] - v;
                    else *(f32x4*)dst[qi] = old[qi] + v;          // EPI_STORE: old == 0
                }
            }
        }
        __syncthreads();          // red is reused by the next round
    }
}

// touches 1/32 of the next op's weights so that they sit in this XCD's L2 when the op starts (the values are summed into `sink`,
// which is never stored: the compiler only has to keep the loads)
__device__ __forceinline__ void pk_warm(const PkStep& nx, int rank, int tid, float& sink) {
    if (nx.kind != PK_CONV || tid < 64) return;           // wave 0 keeps its load queue free for the completion poll
    const size_t n16 = (size_t)nx.ntap * nx.Cin_pad * nx.Cout_pad / 4;
    const f32x4* w = (const f32x4*)nx.w;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)rank * (PK_WAVES - 1) * 64 + (tid - 64); i < n16; i += (size_t)PK_CHUNKS * (PK_WAVES - 1) * 64) s += w[i];
    sink += s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(PK_WAVES * 64) void pk_flow_kernel(PkFlowArgs A_) {
    extern __shared__ __attribute__((aligned(16))) float red[];     // [16 waves][4 quads][64 lanes] x 16 B = 64 KB | op program | op geometry
    __shared__ int s_chunk[2];
    __shared__ PkFlowArgs s_args;
    PkStep* sprog = (PkStep*)(red + (size_t)PK_WAVES * 16 * 64);
    PkGeo* sgeo = (PkGeo*)(sprog + PK_MAX_STEPS);
    const int tid = threadIdx.x;
    const int x = (int)(pk_xcc_id() & 7u);
    // Everything the op loop needs is put into LDS once: the arguments (the compiler would otherwise re-read the kernarg segment
    // inside the loop), the op program, and -- one thread per op -- the op's geometry on this XCD's window (tile counts, waves per
    // tile, chunking: integer divisions that would otherwise sit on every op's critical path)
    if (tid == 0) s_args = A_;
    {
        const int nw = A_.nsteps * (int)(sizeof(PkStep) / 4);
        for (int i = tid; i < nw; i += PK_WAVES * 64) ((int*)sprog)[i] = ((const int*)A_.prog)[i];
    }
    __syncthreads();
    const PkFlowArgs& A = s_args;
    PkWin W;
    W.f0 = x * A.fs; W.f1 = W.f0 + A.fs < A.F ? W.f0 + A.fs : A.F;
    W.fa = W.f0 - A.halo_total > 0 ? W.f0 - A.halo_total : 0;
    W.fb = W.f1 + A.halo_total < A.F ? W.f1 + A.halo_total : A.F;
    const int nsteps = __builtin_amdgcn_readfirstlane(A.nsteps);
    if (tid < nsteps) {
        const PkStep& st = sprog[tid];
        PkGeo ge;
        const int lo_f = W.f0 - st.halo > 0 ? W.f0 - st.halo : 0, hi_f = W.f1 + st.halo < A.F ? W.f1 + st.halo : A.F;
        ge.n_lo = lo_f - W.fa; ge.n_hi = hi_f - W.fa; ge.ncol = 0; ge.nt = 0; ge.per = 0; ge.ks = 1; ge.nch = PK_CHUNKS; ge.pad = 0;
        if (st.kind == PK_CONV) {
            ge.ncol = (ge.n_hi - ge.n_lo + 31) / 32;
            ge.nt = (st.Cout_pad / 32) * ge.ncol;
            // waves per tile: from the WIDEST window of this op (the same on every XCD, so the summation order of a frame does
            // not depend on which XCD computes it): fill the XCD's 512 wave slots, keep >= 3 K groups per wave
            const int wmax = A.fs + 2 * st.halo < A.F ? A.fs + 2 * st.halo : A.F;
            const int ntmax = (st.Cout_pad / 32) * ((wmax + 31) / 32);
            const int gall = st.ntap * (st.Cin_pad / 8);
            int ks = PK_WAVES;
            while (ks > 1 && (ntmax * ks > PK_CHUNKS * PK_WAVES || gall < 3 * ks)) ks >>= 1;
            const int G = PK_WAVES / ks;
            ge.ks = ks;
            ge.per = G * ((ge.nt + PK_CHUNKS * G - 1) / (PK_CHUNKS * G));
            ge.nch = (ge.nt + ge.per - 1) / ge.per;
        }
        sgeo[tid] = ge;
    }
    __syncthreads();
    float sink = 0.f;
    if (W.f0 < A.F) {
        float* priv = A.priv + (size_t)x * A.priv_stride;
        unsigned* claim = A.ctr + (size_t)x * 2 * PK_MAX_STEPS;
        unsigned* done = claim + PK_MAX_STEPS;
        long long* trace = A.trace ? A.trace + (size_t)blockIdx.x * PK_MAX_STEPS * 8 : nullptr;
        unsigned cnext = 0;
        if (tid == 0) {
            cnext = __hip_atomic_fetch_add(&claim[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_chunk[0] = (int)cnext;
        }
        __syncthreads();
        const int rank = s_chunk[0] & (PK_CHUNKS - 1);   // this workgroup's number among those serving the XCD
        int slot = 0;                                     // s_chunk[slot] holds the chunk to process next (published behind a barrier)
        for (int s = 0; s < nsteps; s++) {
            const int kind = __builtin_amdgcn_readfirstlane(sprog[s].kind);
            const int nch = __builtin_amdgcn_readfirstlane(sgeo[s].nch);
            if (trace && tid == 0) trace[s * 8 + 0] = (long long)__builtin_amdgcn_s_memtime();
            unsigned c = cnext;        // (thread 0) the claim made for this op while the previous one ran
            if (tid == 0 && s + 1 < nsteps) cnext = __hip_atomic_fetch_add(&claim[s + 1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool may_reclaim = true, warmed = false;
            for (;;) {
                const int chunk = s_chunk[slot];
                slot ^= 1;
                if (chunk >= 0 && chunk < nch) {
                    if (kind == PK_CONV) pk_conv_chunk(A, sprog[s], sgeo[s], W, priv, chunk, red, tid, trace ? trace + s * 8 : nullptr);
                    else if (kind == PK_EXPAND) pk_expand(A, W, priv + (size_t)sprog[s].out_buf * A.rows * A.wld, chunk, tid);
                    else pk_store_out(A, W, priv + (size_t)sprog[s].in_buf * A.rows * A.wld, chunk, tid);
                    pk_stores_done();
                    if (trace && tid == 0) trace[s * 8 + 7] = (long long)__builtin_amdgcn_s_memtime();
                    __syncthreads();
                    if (tid == 0) __hip_atomic_fetch_add(&done[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (trace && tid == 0) { trace[s * 8 + 1] = (long long)__builtin_amdgcn_s_memtime(); trace[s * 8 + 3] = (long long)(x * 1000 + chunk + 1); }
                } else {
                    // nothing to do for this workgroup (its claim was beyond the op's chunks): thread 0 must not publish the next
                    // chunk before every wave has read this one and the "more" word of the round before.  Without this barrier a
                    // workgroup that starts late and runs through ops that are already complete (several engines sharing the GPU)
                    // let its slower waves read thread 0's NEXT pair of words: a chunk processed twice, an op counted complete
                    // early, one wrong window in ~1 of 50 calls (tools/concurrent_engines_check.py)
                    __syncthreads();
                }
                // while the op completes elsewhere: pull this workgroup's share of the NEXT op's weights into the XCD's L2
                if (!warmed && s + 1 < nsteps) { pk_warm(sprog[s + 1], rank, tid, sink); warmed = true; }
                // wait for the op to complete on this XCD; after a while, look for chunks nobody has claimed (workgroups that
                // are not resident): the kernel never depends on the presence of a particular workgroup.  The same barrier that
                // ends the wait publishes the chunk to process next: another one of this op, or the one claimed for the next op
                if (tid == 0) {
                    int polls = 0, again = -1;
                    while (__hip_atomic_load(&done[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nch) {
                        __builtin_amdgcn_s_sleep(1);
                        if (may_reclaim && ++polls >= 12) {
                            c = __hip_atomic_fetch_add(&claim[s], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (c < (unsigned)nch) { again = (int)c; break; }
                            may_reclaim = false;        // every chunk is claimed: only waiting is left
                        }
                    }
                    s_chunk[slot] = again >= 0 ? again : (int)cnext;
                    s_chunk[slot ^ 1] = again >= 0 ? 1 : 0;          // (the slot just consumed) "stay on this op"
                }
                __syncthreads();
                const bool more = s_chunk[slot ^ 1] != 0;
                if (trace && tid == 0 && !more) trace[s * 8 + 2] = (long long)__builtin_amdgcn_s_memtime();
                if (!more) break;
            }
        }
    }
    // ---- the last workgroup to leave re-arms the counters for the next launch (stream order makes it visible)
    __syncthreads();
    if (tid == 0) s_chunk[0] = __hip_atomic_fetch_add(&A.ctr[8 * 2 * PK_MAX_STEPS], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
    __syncthreads();
    if (s_chunk[0])
        for (int i = tid; i <= 8 * 2 * PK_MAX_STEPS; i += PK_WAVES * 64) __hip_atomic_store(&A.ctr[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (sink == 123.456f) A.ctr[8 * 2 * PK_MAX_STEPS + 1] = 1u;      // never true: keeps the warm-up loads alive
}

size_t pk_counter_bytes() { return (size_t)(8 * 2 * PK_MAX_STEPS + 16) * sizeof(unsigned); }

void pk_flow(const PkFlowArgs& A, hipStream_t st) {
    const size_t lds = (size_t)PK_WAVES * 16 * 64 * sizeof(float) + (size_t)PK_MAX_STEPS * (sizeof(PkStep) + sizeof(PkGeo));      // partial tiles (64 KB) + the op program
    (void)hipFuncSetAttribute((const void*)pk_flow_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);   // per device
    hipLaunchKernelGGL(pk_flow_kernel, dim3(8 * PK_CHUNKS), dim3(PK_WAVES * 64), lds, st, A);
}

}  // namespace sts
#endif  // STS_EXPERIMENTS
