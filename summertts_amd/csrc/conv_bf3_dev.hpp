// conv_bf3_dev.hpp -- device code shared by conv_bf3.hip, conv_bf3_group.hip and resblock_bf3.hip (one translation unit each,
// so that they compile in parallel).
//
// fp32 Conv1d / ConvTranspose1d on the BF16 matrix cores of gfx950 (v_mfma_f32_32x32x16_bf16).
//
// Same contract as conv.hip's kernels (they replace /root/reference/src/nn_op/nn_conv1d.cpp:118-199 and
// nn_conv1d_transposed.cpp:106-150), same ConvArgs, same epilogues -- a different way of doing the fp32 arithmetic.
// gfx950 has no TF32-like mode and its exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the bf16 rate
// (MI355X_MICROARCH.md: 157 TF/s vs 2.5 PF/s).  An fp32 number is the EXACT sum of three bf16 numbers
//     x = hi + mid + lo,   hi = trunc_bf16(x), mid = trunc_bf16(x - hi), lo = x - hi - mid      (8 + 8 + 8 mantissa bits)
// so a product of two fp32 numbers is the sum of nine bf16 x bf16 products, each of which the matrix core forms exactly
// and accumulates in fp32.  The three products of relative order 2^-24 and below (mid*lo, lo*mid, lo*lo) are dropped --
// that is the size of the rounding error an fp32 multiply-add makes anyway -- leaving SIX bf16 MFMAs per fp32 MFMA-equivalent:
// 6/16 of the matrix-pipe time of the exact-fp32 instruction.  Measured against fp64 the result is as accurate as the
// fp32 MFMA kernels' (tests/test_parity_gpu.py::test_bf3_conv_*; docs/HISTORY.md 5d); every parity tolerance is unchanged.
//
// Layout follows from the instruction: a lane feeds 8 CONSECUTIVE k values (input channels) of one row / column.
//   * weights are split and fragment-packed at load time (bf3_pack): [phase][chunk of 16 cin][tap][32-row tile][plane][lane][8],
//     so an A fragment is one 16-byte load per lane, 1 KB contiguous per wave, (step, row tile, plane) in the scalar offset;
//   * the input window of a 16-channel chunk is staged ONCE in LDS, already split, channel-minor: plane[pos][16 cin] bf16
//     (32 B per position and plane; the 16-byte half a lane reads is XOR-swizzled with bit 3 of the position, which
//     makes every ds_read_b128 lane group hit 16 distinct 16-byte bank slots for ANY tap shift).  The transposition
//     (global memory is channel-major, time contiguous) happens in the staging registers: lane (pos, half) loads its 8
//     channels of one position (every load instruction reads two full 128-byte lines), applies the fused input
//     activation, splits, and writes three 16-byte vectors.  The split costs ~7 VALU ops per staged element and is
//     amortised over all output rows and taps that read it;
//   * a wave owns a (32 MW) x (32 NW) output tile: every A fragment is reused over NW column tiles and every B fragment
//     over MW row tiles, 6 MW NW MFMAs per (chunk, tap) step, accumulators interleaved so no MFMA waits for its predecessor.
#pragma once
#include "kernels.hpp"
#include "devmath.hpp"
#include "conv_common.hpp"
#include "knobs.hpp"

#include <stdlib.h>
#include <string.h>

namespace sts {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#ifndef STS_EXP
#define STS_EXP 0   // timing experiments only (tools/exp_build.sh); 0 in every shipped build
#endif
// A/B switches of round 3's instruction-count work on the staged kernel (tools/var_build.sh builds one library per mask; every
// mask computes identical results).  Measured on MI355X at one utterance (profiles/r03_bf3_variants.log; box-to-box spread ~3 %):
//   1  no scheduling barrier in front of a step's MFMAs ............... +0.3 % (slower)
//   2  leaky relu as max(v, slope v) (2 instead of 4 VALU per staged value) \ together -1.4 % of the trunk: kept (default 6)
//   4  plain tiles: step index == position in the packed weights ......... /
//   8  a step's LDS reads / weight loads interleaved with its MFMAs (sched_group_barrier) ... no effect
//   16 weight fragments requested two steps ahead (ring of 3, 214 VGPRs) .................... no effect
//   32 weight loads AND LDS reads of the next step issued between this step's MFMAs (one basic block, 2 MFMA : 1 memory op) ..
//      +7 % SLOWER (a lone workgroup's step 1 500 -> 1 900 cycles: anything placed between MFMAs of one accumulator chain costs
//      more than its issue slot, and the fragments are requested later)
// i.e. the kernel is bound neither by instruction issue in the staging / bookkeeping code nor by weight latency; a lone wave needs
// ~1 500 cycles per 24-MFMA step (768 pipe cycles), two waves per SIMD ~1 650 each = the pipe 94 % busy during K loops
// (tools/tile_trace_conv.py, profiles/r03_tile_trace_conv_kloop_per_step.log): what is left is outside the K loop.
#ifndef STS_VAR
#define STS_VAR 6
#endif
// Lab switches of the two-term fp16 kernels (tools/var_build.sh with VAR_EXTRA / VAR_TAG; every setting computes identical results).
// Measured on MI355X, one utterance / batch 32 (round 3):
//   STS_H2_AR     ring of weight fragments of the plain tiles: step s + AR - 1 is requested during step s.  3 and 4: no effect
//                 (the K loop is not waiting for weights)
//   STS_H2_WAVES / STS_H2_MINW   most / fewest waves per SIMD the register budget is sized for.  3 / 3 (168 registers, three
//                 128 x 128 workgroups per CU): +1 % / -2 %
//   (removed again) steps handled in pairs -- operands of steps s + 2, s + 3 requested, then 24 MFMAs back to back, rings of 4: no effect
// tools/h2_decomp.sh / h2_decomp2.sh (parts of a step compiled out, profiles/r03_f16x2_kloop_decomposition.log): two workgroups per CU
// spend ~1 070 cycles per 12-MFMA step = the pipe 72 % busy in the K loop (split-bf16: 1 650 per 24 = 93 %).  With nothing but the
// MFMAs and the loop bookkeeping left: 760-810 (the pipe's own rate with two waves per SIMD; ONE wave per SIMD needs 632 -- its
// bookkeeping does not overlap its own MFMAs).  The other ~280 cycles are the step's memory operations, none of them dominant:
// weight loads 150, staging loads 80, LDS reads 40, barriers 15-40 -- 0.67 KB of operands per MFMA against split-bf16's 0.5, through
// the same vector-memory and LDS pipes in half the time.  Prefetch depth, occupancy and burst length do not change that; a larger
// tile per wave (fewer operand bytes per MFMA) would, and needs the accumulators in AGPRs at one wave per SIMD: not built.
#ifndef STS_H2_AR
#define STS_H2_AR 2
#endif
// STS_GROUP_RPF (lab switch, default off): the grouped trunk launches request their residual tile before the K loop (236 registers instead of
// 172, still two waves per SIMD).  Measured in round 4 (profiles/r04_ab_log.md): decoder 1.783-1.787 vs 1.765-1.772 ms at one utterance, a tie at
// batch 32 -- the 64 early 16-byte loads sit in front of the first weight fragments (loads return in order) and cost the prologue what they
// save the epilogue.
#ifndef STS_GROUP_RPF
#define STS_GROUP_RPF 0
#endif
// Occupancy of the staged two-term kernels (round 5, profiles/r05_ab_log.md session 5).  STS_MW1_WAVES: the single-conv tiles whose waves own 32 rows need 124-148
// registers by themselves; allowing a THIRD workgroup per CU turns the last upsampler of one utterance (669 workgroups on 512 slots: a full round and a
// 30 % one, profiles/r04_tile_trace.log launch 34) into one round: 35.1 -> 25.6 us.  STS_GROUP_MINW / _WAVES: the plain 128 x 128 tile of the GROUPED launches
// (the 128-channel ResBlock stage) capped at 168 registers = three workgroups per CU (a 20-byte spill): 104.0 -> 99.4 us per launch at one utterance,
// MB-iSTFT batch 64 -3.2 % of the step.  The same cap on every staged kernel is a wash: the phase-merged upsampler tile that forms the chain mean while
// staging (198 registers) spills and doubles (34 -> 64 us), which is why the switches are per kernel family.
#ifndef STS_MW1_WAVES
#define STS_MW1_WAVES 3
#endif
#ifndef STS_GROUP_MINW
#define STS_GROUP_MINW 3
#endif
#ifndef STS_GROUP_WAVES
#define STS_GROUP_WAVES 3
#endif
#ifndef STS_H2_MINW
#define STS_H2_MINW 1
#endif
#ifndef STS_H2_WAVES
#define STS_H2_WAVES 2
#endif

#ifdef STS_TILE_TRACE
// Lab build only (tools/var_build.sh with -DSTS_TILE_TRACE): every workgroup of the staged / fused kernels appends one 12-word record
// {gridDim.x, blockIdx.x, kind | HW_ID << 8 | XCC_ID << 40, realtime (100 MHz) at start, s_memtime stamps [6], realtime at end, group member}.
// Word 0 of the buffer is the record counter, records start at word 16.  The kernels live in three translation units; each keeps
// its own copy of the buffer address (no relocatable device code), sts_debug_tile_trace (conv_bf3.hip) binds all three.
constexpr int TT_WORDS = 12, TT_HEAD = 16;
static __device__ long long* g_tile_trace = nullptr;
static __device__ unsigned g_tile_trace_cap = 0;
static inline int tile_trace_bind(long long* buf, unsigned capacity_records) {
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace), &buf, sizeof(buf)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace_cap), &capacity_records, sizeof(unsigned)) != hipSuccess) return -1;
    return 0;
}
__device__ __forceinline__ long long* tt_open(int kind, int member) {
    if (!g_tile_trace || threadIdx.x != 0) return nullptr;
    const unsigned slot = atomicAdd((unsigned*)g_tile_trace, 1u);
    if (slot >= g_tile_trace_cap) return nullptr;
    long long* r = g_tile_trace + TT_HEAD + (size_t)slot * TT_WORDS;
    r[0] = (long long)gridDim.x; r[1] = (long long)blockIdx.x;
    r[2] = (long long)kind | ((long long)(__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4)) & 0xffffffffll) << 8 | (long long)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xf) << 40;
    r[3] = (long long)__builtin_amdgcn_s_memrealtime();
    r[11] = (long long)member;
    return r;
}
#define TT_STAMP(i) do { if (tt_rec) tt_rec[4 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define TT_CLOSE() do { if (tt_rec) tt_rec[10] = (long long)__builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TT_STAMP(i) do { } while (0)
#define TT_CLOSE() do { } while (0)
#endif

// exact three-way split of 8 fp32 values (one lane's 8 channels) into bf16 planes; element e of a plane sits in the low
// (e even) / high (e odd) half of dword e / 2 -- the order v_mfma_*_bf16 reads its 8 k values in
__device__ __forceinline__ void split8(const float (&x)[8], u32x4& hi, u32x4& mid, u32x4& lo) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const unsigned ua = __builtin_bit_cast(unsigned, x[2 * d]), ub = __builtin_bit_cast(unsigned, x[2 * d + 1]);
        hi[d] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
        const float ra = x[2 * d] - __builtin_bit_cast(float, ua & 0xffff0000u);
        const float rb = x[2 * d + 1] - __builtin_bit_cast(float, ub & 0xffff0000u);
        const unsigned va = __builtin_bit_cast(unsigned, ra), vb = __builtin_bit_cast(unsigned, rb);
        mid[d] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
        const float la = ra - __builtin_bit_cast(float, va & 0xffff0000u);
        const float lb = rb - __builtin_bit_cast(float, vb & 0xffff0000u);
        lo[d] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, lb), __builtin_bit_cast(unsigned, la), 0x07060302u);
    }
}

__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// the six products, smallest terms first: (lo,hi) (hi,lo) (mid,mid) (mid,hi) (hi,mid) (hi,hi); planes 0 / 1 / 2 = hi / mid / lo
constexpr int kProdA[6] = {2, 0, 1, 1, 0, 0};
constexpr int kProdB[6] = {0, 2, 1, 0, 1, 0};

// ------------------------------------------------------------------------------------------------
// MATH 1 ("f16x2", round 3): the same kernels with every fp32 operand as TWO fp16 terms and THREE products per fp32 product --
// half the matrix-pipe time of the split-bf16 form.  fp16 carries 11 significant bits, so hi + lo holds 22-23 of an fp32's 24;
// what makes it usable is keeping the small term out of fp16's subnormal range:
//   activation x:  hi = fp16(x),  lo' = fp16((x - hi) * 2^11)        (the residual is exact in fp32; scaled it is as large as x)
//   weight     w:  ws = w * 2^s with max |ws| in [2^13, 2^14) per conv (bf3_pack math 1, host);  P0 = fp16(ws), P1 = fp16(ws - P0)
//                  are packed (two planes: 2 KB per 32-row tile and step instead of 3 KB);  P2 = P0 * 2^-11 (exact above the
//                  subnormals) costs the kernel one packed multiply per fragment dword -- a third less weight traffic out of L2
//   x * ws  ~=  hi * P0  +  hi * P1  +  lo' * P2        (dropped: lo * lo, relative 2^-22 worst case, ~2^-24.6 rms)
// every product is exact in the fp32 accumulator; the tile is multiplied by 2^-s before the epilogue.  |x| > 65504 does not fit
// fp16: the staging code tracks max |x| and raises ConvArgs::ovf, the engine then repeats the utterance in the split-bf16 form.
// ------------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr float kH2Limit = 60000.f;
#ifndef STS_SPLIT_MIX
#define STS_SPLIT_MIX 1        // lab: 0 = the residual through convert-back / subtract / scale / convert (rounds 3-5; the same bits)
#endif
__device__ __forceinline__ void split8h(const float (&x)[8], u32x4& hi, u32x4& lo, float& amax) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const f32x2 v = {x[2 * d], x[2 * d + 1]};
        const f16x2 h = __builtin_convertvector(v, f16x2);                 // round to nearest even
#if STS_SPLIT_MIX
        // lo' = fp16(2048 x - 2048 hi) as ONE mixed-precision FMA per value (hi read as fp16, the sum is exact in fp32, one rounding to fp16):
        // the same bits as the form below at 4 instead of 7 VALU instructions per pair (round 6)
        const f32x2 v2 = v * 2048.f;
        const unsigned hu = __builtin_bit_cast(unsigned, h);
        unsigned l = 0u;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "+v"(l) : "v"(hu), "s"(-2048.0f), "v"(v2[0]), "v"(v2[1]));
        hi[d] = hu;
        lo[d] = l;
#else
        const f32x2 r = (v - __builtin_convertvector(h, f32x2)) * 2048.f;
        const f16x2 l = __builtin_convertvector(r, f16x2);
        hi[d] = __builtin_bit_cast(unsigned, h);
        lo[d] = __builtin_bit_cast(unsigned, l);
#endif
        amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])));
    }
}
__device__ __forceinline__ f32x16 mfma_f16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
// weight planes P2 / P1 / P0 against activation planes lo' / hi / hi, smallest term first
constexpr int kProdAh[3] = {2, 1, 0};
constexpr int kProdBh[3] = {1, 0, 0};

// one (chunk, tap) step of a wave's tile in either arithmetic
template <int MATH, int MW, int NW, int NPA, int NPB>
__device__ __forceinline__ void step_mfmas(f32x16 (&acc)[MW][NW], const u32x4 (&ac)[MW][NPA], const u32x4 (&bc)[NW][NPB]) {
    if constexpr (MATH == 0) {
#pragma unroll
        for (int p = 0; p < 6; p++)
#pragma unroll
            for (int i = 0; i < MW; i++)
#pragma unroll
                for (int q = 0; q < NW; q++) acc[i][q] = mfma_bf16(ac[i][kProdA[p]], bc[q][kProdB[p]], acc[i][q]);
    } else {
        u32x4 p2[MW];                               // P2 = P0 * 2^-11
#pragma unroll
        for (int i = 0; i < MW; i++) p2[i] = __builtin_bit_cast(u32x4, __builtin_bit_cast(f16x8, ac[i][0]) * (_Float16)0.00048828125f);
#ifndef STS_H2_PRODUCTS_FROM
#define STS_H2_PRODUCTS_FROM 0      // lab (WRONG results): 1 = only two of the three products -- what would a third fewer MFMAs buy? (the Winograd question)
#endif
#pragma unroll
        for (int p = STS_H2_PRODUCTS_FROM; p < 3; p++)
#pragma unroll
            for (int i = 0; i < MW; i++)
#pragma unroll
                for (int q = 0; q < NW; q++) acc[i][q] = mfma_f16(p == 0 ? p2[i] : ac[i][kProdAh[p]], bc[q][kProdBh[p]], acc[i][q]);
    }
}


template <int MW, int NW, int WM, int WN, int NSUB = 1, int KG = 1, bool NTL = false, int MATH = 0, bool NSUM = false, bool RPF = false>
__device__ __forceinline__ void conv_bf3_body(const ConvArgs& a, const int mtiles, const int bx, const int by, const int b, const int pm = 0, [[maybe_unused]] const int tt_member = 0) {
    // pm (polyphase transposed convs): the workgroup's MT rows run over the MERGED row space phase * Cout_pad + row, so that
    // several phases (or all row blocks of a phase) share ONE staged, split input window instead of staging it once each
    // NSUB: 16-channel sub-chunks staged per barrier (a staged chunk = 16 NSUB channels): fewer barriers and more bytes in
    // flight per workgroup for the few-tap convs, at NSUB x the staging registers and LDS
    // KG: wave groups that split K inside the workgroup (a grid-starved conv with a long K loop: the 256-channel stage of one
    // utterance has only 252 tiles of 128 x 128): group g owns the sub-chunks g, g + KG, ... of every staged chunk; the
    // partial tiles are exchanged through LDS once, each group then finishes the column tiles q = g (mod KG)
    static_assert(NSUB % KG == 0 && (KG == 1 || NW % KG == 0), "K groups take whole sub-chunks and whole column tiles");
    constexpr int MT = 32 * MW * WM, NT = 32 * NW * WN, NTW = WM * WN, NWAVE = NTW * KG;
    constexpr int WIN = NT + MAX_HALO;                 // staged positions per chunk
    constexpr int NSLOT = WIN / 32;                    // staging slots of 32 positions x 16 channels (one wave-wide load group)
    constexpr int NITEM = NSLOT * NSUB;                // (sub-chunk, slot) items per staged chunk
    constexpr int SPW = (NITEM + NWAVE - 1) / NWAVE;   // items per wave
    constexpr int NPB = MATH ? 2 : 3;                  // planes of a staged activation
    constexpr int NPA = MATH ? 2 : 3;                  // packed planes of a weight
    constexpr unsigned ABLK = NPA * 1024u;             // bytes of one (step, 32-row tile) block of the packed weights
    constexpr int PLANE = WIN * 32, SUB = NPB * PLANE, BUF = NSUB * SUB;   // bytes
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
#ifdef STS_TILE_TRACE
    long long* tt_rec = tt_open(0, tt_member);     // kind 0 = staged conv (stamps: start, first barrier, K loop done, epilogue done)
    TT_STAMP(0);
#endif
    // (in_reflect, MB-iSTFT subband conv: the logical input is the reflect-pad-left-1 view of x -- position p reads x[p - 1], p = 0 reads x[1])
    const int orig_len = uni(seg_len(a.in_seg, b));
    const int in_len = orig_len + (a.in_reflect ? 1 : 0);
    const int out_len = uni(seg_len(a.out_seg, b));
    const int n_count = a.transposed ? in_len + a.n_extra : out_len;
    const int n0 = bx * NT;
    if (n0 >= n_count) return;
    const int phase0 = pm ? 0 : by / mtiles;
    const int m0 = pm ? by * MT : (by - phase0 * mtiles) * MT;
    const size_t in_base = (size_t)uni(seg_start(a.in_seg, b)), out_base = (size_t)uni(seg_start(a.out_seg, b));
    const float* const xbase = uni(a.x);
    const long x_ld = uni(a.x_ld);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const int kg = swave / NTW;                                    // scalar
    const int tw = wave - kg * NTW;
    const int wm = tw / WN, wn = tw % WN;
    const int l31 = lane & 31, half = lane >> 5;

    const int first = a.tap_off, last = a.tap_off + (a.ntap - 1) * a.tap_step;
    const int lo = first < last ? first : last, hi = first < last ? last : first;
    const int W = NT + (hi - lo);
    const int win0 = n0 + lo;
    int mbase = m0 + wm * MW * 32;
    int phase = phase0;
    bool wvalid = true;
    if (pm) {                                   // this wave's rows in the merged space -> (phase, row inside the phase)
        phase = mbase / a.Cout_pad;
        mbase -= phase * a.Cout_pad;
        wvalid = phase < a.out_stride;
    }

    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; i++)
#pragma unroll
        for (int j = 0; j < NW; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nchunk = a.Cin_pad / (CK * NSUB);
    const int nsteps = nchunk * (NSUB / KG) * a.ntap;          // steps of ONE wave
    const int nsteps_all = nchunk * NSUB * a.ntap;             // 16-channel x tap blocks of the packed weights
    const int nrt = a.Cout_pad / 32;

    // ---- A fragments: step s = (16-channel chunk) * ntap + tap is one contiguous block of nrt * 3 KB
    const rsrc_t wrs = make_rsrc(a.wb3, (unsigned)((size_t)(a.transposed && !a.rowph ? a.out_stride : 1) * nsteps_all * nrt * ABLK));
    // (the wave's row tile goes into the per-lane offset: the compiler cannot prove tid >> 6 wave-uniform and would wrap
    // every load in a readfirstlane loop if it sat in the scalar offset)
    const unsigned a_voff = wvalid ? (unsigned)lane * 16u + ((unsigned)phase * (unsigned)nsteps_all * (unsigned)nrt + (unsigned)(mbase >> 5)) * ABLK : kOOB;
    const unsigned a_s0 = 0u;
    const unsigned a_step = (unsigned)nrt * ABLK;
    auto load_a = [&](int s, u32x4 (&dst)[MW][NPA]) {
        const unsigned sb = a_s0 + (unsigned)s * a_step;
#pragma unroll
        for (int i = 0; i < MW; i++)
#pragma unroll
            for (int pl = 0; pl < NPA; pl++)
                dst[i][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)a_voff, (int)(sb + (unsigned)i * ABLK + (unsigned)(pl * 1024)), 0));
    };
    // ---- B fragments of (sub-chunk, tap j) out of the staged, split window
    const int b_t0 = wn * NW * 32 + l31 + a.tap_off - lo;
    auto load_b = [&](int bufi, int sub, int j, u32x4 (&dst)[NW][NPB]) {
        const int t = b_t0 + j * a.tap_step;
        const unsigned char* sb = smem3 + bufi * BUF + sub * SUB + t * 32 + ((half ^ ((t >> 3) & 1)) << 4);
#pragma unroll
        for (int q = 0; q < NW; q++)
#pragma unroll
            for (int pl = 0; pl < NPB; pl++) dst[q][pl] = *(const u32x4*)(sb + pl * PLANE + q * 1024);
    };

    // ---- input staging: wave w owns items w, w + NWAVE, ... of a chunk; item = (sub-chunk, slot); lane (l31, half) of an item
    // holds channels 8 half .. 8 half + 7 of window position 32 slot + l31.  Raw values wait in registers (chunk c + 1 during
    // chunk c); activation + split at store time.
    unsigned xoff[SPW]; int lds_w[SPW]; int isub[SPW]; bool sact[SPW];
    const unsigned ld4 = (unsigned)a.x_ld * 4u;
#pragma unroll
    for (int i = 0; i < SPW; i++) {
        const int it = swave + i * NWAVE;                           // wave-uniform
        const int sub = it / NSLOT, slot = it - sub * NSLOT;
        const int col = slot * 32 + l31;
        const int pos = win0 + col;
        isub[i] = sub;
        sact[i] = it < NITEM && slot * 32 < W;
        bool v = col < W && pos >= 0 && pos < in_len;
        int src = pos;
        if (a.in_reflect) { src = pos - 1; if (src < 0) { src = 1; v = v && orig_len > 1; } }
        xoff[i] = v ? (unsigned)half * 8u * ld4 + (unsigned)src * 4u : kOOB;
        lds_w[i] = sub * SUB + col * 32 + ((half ^ ((col >> 3) & 1)) << 4);
    }
    float xr[SPW][8];
    // NSUM: the logical input is the mean of 2 or 3 tensors of identical geometry (ConvArgs::nsum): their raw values wait in registers
    // next to x's and are combined at store time, ((x + xs1) + xs2) / nsum in exactly sum_scale's order
    float xr1[NSUM ? SPW : 1][8], xr2[NSUM ? SPW : 1][8];
    const float* const xbase1 = NSUM ? uni(a.xs1) : nullptr;
    const float* const xbase2 = NSUM ? uni(a.xs2) : nullptr;
    const int nsum = NSUM ? uni(a.nsum) : 0;
    int as = 0;        // next A step to request
    const float act_slope = a.in_act ? a.in_slope : 1.0f;
    float amax = 0.f;  // MATH 1: largest staged magnitude this lane has seen
    auto load_x = [&](int c) {
        if ((STS_EXP & 1) && c > 0) return;
        // one descriptor per 16-channel sub-chunk, based at its first row: rows ride in the scalar offset, the per-lane
        // offset (row half + position) is range-checked by the hardware
#pragma unroll
        for (int i = 0; i < SPW; i++)
            if (sact[i]) {
                const size_t row0 = (size_t)(c * NSUB + isub[i]) * CK * x_ld + in_base;
                const unsigned span = (unsigned)((15ul * x_ld + orig_len) * 4ul);
                const rsrc_t rs = make_rsrc(xbase + row0, span);
#pragma unroll
                for (int e = 0; e < 8; e++)
                    xr[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)xoff[i], (int)((unsigned)e * ld4), NTL ? 2 : 0));
                if constexpr (NSUM) {
                    const rsrc_t rs1 = make_rsrc(xbase1 + row0, span);
#pragma unroll
                    for (int e = 0; e < 8; e++)
                        xr1[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs1, (int)xoff[i], (int)((unsigned)e * ld4), 0));
                    if (nsum > 2) {
                        const rsrc_t rs2 = make_rsrc(xbase2 + row0, span);
#pragma unroll
                        for (int e = 0; e < 8; e++)
                            xr2[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs2, (int)xoff[i], (int)((unsigned)e * ld4), 0));
                    }
                }
            }
    };
    auto store_tile = [&](int bufi) {
        if ((STS_EXP & 8) && bufi >= 0 && as > 2) return;
        unsigned char* sb = smem3 + bufi * BUF;
#pragma unroll
        for (int i = 0; i < SPW; i++)
            if (sact[i]) {
                float v[8];
                if constexpr (NSUM) {
                    const float div = (float)nsum;
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        float t = xr[i][e] + xr1[i][e];
                        if (nsum > 2) t += xr2[i][e];
                        xr[i][e] = t / div;
                    }
                }
                if (STS_VAR & 2) {
                    // leaky relu (0 <= slope <= 1) as max(v, slope v): two instructions per value instead of compare + multiply + two selects
                    // (no activation: slope 1).  v < 0: slope v >= v; v >= 0: v >= slope v; -0 / +0 as the select form gives them
#pragma unroll
                    for (int e = 0; e < 8; e++) v[e] = __builtin_fmaxf(xr[i][e], xr[i][e] * act_slope);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; e++) { v[e] = xr[i][e]; if (a.in_act) v[e] = v[e] < 0.f ? v[e] * a.in_slope : v[e]; }
                }
                if constexpr (MATH == 0) {
                    u32x4 ph, pm, pl;
                    split8(v, ph, pm, pl);
                    *(u32x4*)(sb + lds_w[i]) = ph;
                    *(u32x4*)(sb + PLANE + lds_w[i]) = pm;
                    *(u32x4*)(sb + 2 * PLANE + lds_w[i]) = pl;
                } else {
                    u32x4 ph, pl;
                    split8h(v, ph, pl, amax);
                    *(u32x4*)(sb + lds_w[i]) = ph;
                    *(u32x4*)(sb + PLANE + lds_w[i]) = pl;
                }
            }
    };

    // ---- main loop over this wave's steps (chunk, sub-chunk, tap): A (L2) and B (LDS) fragments one step ahead
    // A ring: 2 = the fragments of step s + 1 are requested during step s; 3 (STS_VAR & 16, plain tiles only) = two steps ahead
    constexpr int AR = (NSUB == 1 && KG == 1) ? (MATH == 1 ? STS_H2_AR : ((STS_VAR & 16) ? 3 : 2)) : 2;
    u32x4 fa[AR][MW][NPA], fb[2][NW][NPB];
    int sj = 0, ssub = kg, sc = 0;
    auto a_index = [&](int c, int sub, int j) { return (c * NSUB + sub) * a.ntap + j; };
    auto do_step = [&](u32x4 (&acur)[MW][NPA], u32x4 (&anew)[MW][NPA], u32x4 (&bcur)[NW][NPB], u32x4 (&bnxt)[NW][NPB], int s) {
        int nj = sj + 1, nsub = ssub, nc = sc;
        const bool late_a = (STS_VAR & 32) != 0;       // request the next step's weight fragments in the same block as the MFMAs
        int a_next;
        if ((STS_VAR & (4 | 16 | 32)) && NSUB == 1 && KG == 1) {
            // one sub-chunk, one wave group: the step index IS the position in the packed weights, only (tap, chunk) are tracked
            if (nj == a.ntap) { nj = 0; nc = sc + 1; }
            nsub = 0;
            a_next = s + AR - 1;
        } else {
            if (nj == a.ntap) { nj = 0; nsub = ssub + KG; if (nsub >= NSUB) { nsub = kg; nc = sc + 1; } }
            a_next = a_index(nc, nsub, nj);
        }
        if (!late_a && (!(STS_EXP & 2) || s < 2)) load_a(a_next, anew);   // unconditional: past the last step it reads 0 beyond the descriptor, never used
        if (nc != sc && s + 1 < nsteps) {
            store_tile(nc & 1);           // chunk nc's tile (in registers since the start of chunk sc)
            if (!(STS_EXP & 4)) __syncthreads();              // tile nc visible; everyone is done reading the buffer it replaces
            if (nc + 1 < nchunk) load_x(nc + 1);
        }
        if (late_a) load_a(a_next, anew);
        if (!(STS_EXP & 16) || s < 2) load_b(nc & 1, nsub, nj, bnxt);   // past the last step: stale LDS inside the tile, never used
        if (!(STS_VAR & (1 | 32))) __builtin_amdgcn_sched_barrier(0);
        if (!(STS_EXP & 64)) step_mfmas<MATH, MW, NW, NPA, NPB>(acc, acur, bcur);
        if (STS_VAR & (8 | 32)) {
            // the step's 6 LDS reads and 6 weight loads spread between its MFMAs (2 MFMAs per memory operation)
#pragma unroll
            for (int r = 0; r < 6 * MW * NW / 4; r++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);      // 1 VMEM read
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);      // 2 MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
            }
        }
        sj = nj; ssub = nsub; sc = nc;
    };
    // RPF (grouped trunk launches): the residual of the output tile is requested before the K loop (see tile_res_prefetch)
    f32x4u rpre[RPF ? MW : 1][RPF ? NW : 1][4];
    const bool rpf = RPF && KG == 1 && a.epi == EPI_RESADD && a.out_stride == 1 && wvalid;
    if constexpr (RPF) {
        if (rpf) tile_res_prefetch<MW, NW, NTL>(a, rpre, mbase, n0 + wn * NW * 32, l31, half, n_count, out_len, out_base, phase);
    }
    load_x(0);
    load_a(a_index(0, kg, 0), fa[0]);
    if constexpr (AR >= 3) load_a(1, fa[1]);
    if constexpr (AR >= 4) load_a(2, fa[2]);
    store_tile(0);
    __syncthreads();
    TT_STAMP(1);
    load_b(0, kg, 0, fb[0]);
    if (nchunk > 1) load_x(1);
    for (int s = 0; s < nsteps; s += 2 * AR)
        static_for<0, 2 * AR>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if (u < 2 || s + u < nsteps) { if (s + u < nsteps) do_step(fa[u % AR], fa[(u + AR - 1) % AR], fb[u % 2], fb[(u + 1) % 2], s + u); }
        });

    if constexpr (MATH == 1) {
        if (amax > kH2Limit && a.ovf) *a.ovf = 1u;      // a staged value does not fit fp16: the caller repeats the run in the split-bf16 form
        const float ws = a.wscale;                      // 2^-s of the weight pack
#pragma unroll
        for (int i = 0; i < MW; i++)
#pragma unroll
            for (int j = 0; j < NW; j++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[i][j][r] *= ws;
    }
    if constexpr (KG > 1) {
        TT_STAMP(2);
        // ---- exchange the partial tiles: group g gives away its sums for the column tiles it does not finish
        constexpr int REG = NTW * MW * (NW / KG) * 16 * 64;        // floats per owner region
        // (the launcher sizes the LDS allocation for max(staging buffers, exchange): bf3_lds_bytes)
        float* red = (float*)smem3;
        __syncthreads();                                           // every wave is done with the staged tiles
        static_for<0, MW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            static_for<0, NW>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                constexpr int owner = q % KG;
                if (kg != owner) {
#pragma unroll
                    for (int r = 0; r < 16; r++)
                        red[(size_t)owner * REG + (((size_t)(tw * MW + i) * (NW / KG) + q / KG) * 16 + r) * 64 + lane] = acc[i][q][r];
                }
            });
        });
        __syncthreads();
        static_for<0, KG>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            if (kg == g) {
                f32x16 mine[MW][NW / KG];
                static_for<0, MW>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    static_for<0, NW / KG>([&](auto qc) {
                        constexpr int qq = decltype(qc)::value;
#pragma unroll
                        for (int r = 0; r < 16; r++)
                            mine[i][qq][r] = acc[i][qq * KG + g][r] + red[(size_t)g * REG + (((size_t)(tw * MW + i) * (NW / KG) + qq) * 16 + r) * 64 + lane];
                    });
                });
                static_assert(KG == 1 || NW / KG == 1, "one column tile per group");
                if (wvalid) tile_epilogue<MW, NW / KG>(a, mine, mbase, n0 + wn * NW * 32 + g * 32, l31, half, n_count, out_len, out_base, phase, b);
            }
        });
#ifdef STS_TILE_TRACE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TT_STAMP(3);
        TT_CLOSE();
#endif
        return;
    }
    if ((STS_EXP & 32) && acc[0][0][0] != 12345.f) return;
    TT_STAMP(2);
    if constexpr (RPF) {
        if (wvalid) tile_epilogue_pre<MW, NW, NTL>(a, acc, mbase, n0 + wn * NW * 32, l31, half, n_count, out_len, out_base, phase, b, rpre, rpf);
    } else {
        if (wvalid) tile_epilogue<MW, NW, NTL>(a, acc, mbase, n0 + wn * NW * 32, l31, half, n_count, out_len, out_base, phase, b);
    }
#ifdef STS_TILE_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stores have left
    TT_STAMP(3);
    TT_CLOSE();
#endif
}

template <int MW, int NW, int WM, int WN, int NSUB, int KG = 1, int MATH = 0, bool NSUM = false>
__global__ __launch_bounds__(WM* WN * KG * 64) __attribute__((amdgpu_waves_per_eu(MATH ? STS_H2_MINW : 1, MATH ? (MW == 1 && KG == 1 ? STS_MW1_WAVES : STS_H2_WAVES) : 2))) void conv_bf3_kernel(ConvArgs a, int mtiles, int nx, int ny, int pm) {
    const TileId t = map_tile(nx, ny, a.B);
    if (!t.valid) return;
    conv_bf3_body<MW, NW, WM, WN, NSUB, KG, false, MATH, NSUM>(a, mtiles, t.bx, t.by, t.bz, pm);
}

// grouped launch (layer d of all ResBlock chains of a stage in one grid), see conv_mfma_group_kernel
template <int MW, int NW, int WM, int WN, int NSUB, int KG = 1, int MATH = 0>
__global__ __launch_bounds__(WM* WN * KG * 64) __attribute__((amdgpu_waves_per_eu(MATH ? (KG == 1 && MW == 2 && WM == 2 ? STS_GROUP_MINW : STS_H2_MINW) : 1,
                                                                                     MATH ? (KG == 1 && MW == 2 && WM == 2 ? STS_GROUP_WAVES : STS_H2_WAVES) : 2))) void conv_bf3_group_kernel(ConvGroup G, int mtiles, int B, int nx, int ny, int interleave) {
    const TileId t = map_tile(nx, ny, B * G.n);
    if (!t.valid) return;
    // interleave: consecutive dispatch units belong to different members (different K lengths), so that workgroups that
    // share a CU do not run their load / MFMA / store phases in lockstep
    int gi, bx, b;
    if (interleave) { const int unit = t.bz * nx + t.bx; gi = unit % G.n; const int rest = unit / G.n; bx = rest % nx; b = rest / nx; }
    else { gi = t.bz / B; bx = t.bx; b = t.bz - gi * B; }
    const ConvArgs* ga = (const ConvArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    conv_bf3_body<MW, NW, WM, WN, NSUB, KG, false, MATH, false, STS_GROUP_RPF && KG == 1 && MATH == 1>(ga[gi], mtiles, bx, t.by, b, 0, gi);
}

// tile codes: t = 0..5 below with 16-channel chunks, 8 + t the same tile with 32-channel chunks (NSUB = 2)
//   0: 128 x 128 (4 waves of 64 x 64)   1: 64 x 256 (4 waves)   2: 128 x 256 (8 waves)   3: 64 x 128 (2 waves)
//   4:  32 x 256 (4 waves of 32 x 64)   5: 32 x 128 (2 waves)
constexpr int kNumBf3Tiles = 8;      // 6 / 7 (round 3): a wave owns 32 rows x 128 columns -- no two waves of a workgroup fetch the same weight rows
[[maybe_unused]] static bool h2_tile(int tile) { return tile == 0 || tile == 3 || tile == 4 || tile == 20 || tile == 22 || tile == 23 || tile == 24; }   // built for MATH 1
// tile codes this build carries: the ones the automatic choice uses; every other code only in the lab build (-DSTS_EXPERIMENTS)
[[maybe_unused]] static bool bf3_tile_ok(int tile) {
#ifdef STS_EXPERIMENTS
    return (tile >= 0 && tile < kNumBf3Tiles) || (tile >= 8 && tile < 8 + kNumBf3Tiles) || (tile >= 20 && tile < 25);
#else
    return tile == 0 || tile == 3 || tile == 4 || tile == 20 || tile == 22 || tile == 23;
#endif
}

[[maybe_unused]] static int pick_bf3_tile(int Cout_pad, long max_n, long units, bool transposed = false, int math = 0) {
    // units = utterances x group members x phases
    // (a polyphase transposed conv stages the same window once per phase and row block: always the tallest tile)
    if (transposed) {
        // (few tiles and a long K -- HiFi-GAN's first upsampler at one utterance: 6 x 2 x 8 tiles, K = 2 x 512 -- : K split over two
        // wave groups; narrow outputs: the phases share one workgroup's staged window, tile codes 22 / 23; measured per shape in
        // profiles/r02_bf3_conv_microbench.log)
        if (Cout_pad % 128 == 0) return (max_n + 127) / 128 * (Cout_pad / 128) * units < 256 ? 20 : 0;
        return Cout_pad % 64 == 0 ? 22 : 23;
    }
    if (Cout_pad % 128 == 0) {
        const long n128 = (max_n + 127) / 128 * (Cout_pad / 128) * units;
        // two-term fp16: with the matrix time halved, a grid of about one 128 x 128 tile per CU (the 256-channel stage of one
        // utterance: 252) does better as 8-wave workgroups that split K between two wave groups (two waves per SIMD from one
        // workgroup) than as 504 four-wave workgroups of 32 x 256: -3 % of the trunk (STS_BF3_GROUP_TILE sweep, round 3)
        if (math == 1 && n128 >= 192 && n128 < 512) return 20;
        return n128 >= 512 ? 0 : 4;
    }
    if (Cout_pad % 64 == 0) return 3;
    return 4;
}

template <int MW, int NW, int WM, int WN, int NSUB, int KG, int MATH = 0>
static constexpr size_t bf3_lds_bytes() {
    constexpr size_t stage = (size_t)(MATH ? 4 : 6) * NSUB * (32 * NW * WN + MAX_HALO) * 32;
    constexpr size_t xchg = KG > 1 ? (size_t)KG * WM * WN * MW * (NW / KG) * 16 * 64 * 4 : 0;     // partial tiles of the K groups
    return stage > xchg ? stage : xchg;
}

}  // namespace sts
