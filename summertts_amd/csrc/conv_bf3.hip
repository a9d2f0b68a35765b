// conv_bf3.hip -- fp32 Conv1d / ConvTranspose1d on the BF16 matrix cores of gfx950 (v_mfma_f32_32x32x16_bf16).
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
// fp32 MFMA kernels' (tests/test_parity_gpu.py::test_bf3_conv_*); every parity tolerance is unchanged.
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
#ifndef STS_H2_MINW
#define STS_H2_MINW 1
#endif
#ifndef STS_H2_WAVES
#define STS_H2_WAVES 2
#endif

#ifdef STS_TILE_TRACE
// Lab build only (tools/var_build.sh with -DSTS_TILE_TRACE): every workgroup of the staged kernel appends one record
// {gridDim.x, blockIdx.x, xcc | hw id, realtime at start, s_memtime at start / first barrier / K loop done / epilogue done}
__device__ long long* g_tile_trace = nullptr;
__device__ unsigned g_tile_trace_cap = 0;
__device__ unsigned g_tile_trace_n = 0;
extern "C" int sts_debug_tile_trace(long long* buf, unsigned capacity_records) {
    unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace), &buf, sizeof(buf)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace_cap), &capacity_records, sizeof(unsigned)) != hipSuccess) return -1;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_tile_trace_n), &zero, sizeof(unsigned)) != hipSuccess) return -1;
    return 0;
}
extern "C" int sts_debug_tile_trace_count() {
    unsigned n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tile_trace_n), sizeof(unsigned)) != hipSuccess) return -1;
    return (int)n;
}
#define TT_STAMP(i) do { if (tt_rec && threadIdx.x == 0) tt_rec[4 + (i)] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define TT_STAMP(i) do { } while (0)
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
__device__ __forceinline__ void split8h(const float (&x)[8], u32x4& hi, u32x4& lo, float& amax) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const f32x2 v = {x[2 * d], x[2 * d + 1]};
        const f16x2 h = __builtin_convertvector(v, f16x2);                 // round to nearest even
        const f32x2 r = (v - __builtin_convertvector(h, f32x2)) * 2048.f;
        const f16x2 l = __builtin_convertvector(r, f16x2);
        hi[d] = __builtin_bit_cast(unsigned, h);
        lo[d] = __builtin_bit_cast(unsigned, l);
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
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int i = 0; i < MW; i++)
#pragma unroll
                for (int q = 0; q < NW; q++) acc[i][q] = mfma_f16(p == 0 ? p2[i] : ac[i][kProdAh[p]], bc[q][kProdBh[p]], acc[i][q]);
    }
}


template <int MW, int NW, int WM, int WN, int NSUB = 1, int KG = 1, bool NTL = false, int MATH = 0>
__device__ __forceinline__ void conv_bf3_body(const ConvArgs& a, const int mtiles, const int bx, const int by, const int b, const int pm = 0) {
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
    long long* tt_rec = nullptr;
    if (g_tile_trace && threadIdx.x == 0) {
        const unsigned slot = atomicAdd(&g_tile_trace_n, 1u);
        if (slot < g_tile_trace_cap) {
            tt_rec = g_tile_trace + (size_t)slot * 10;
            tt_rec[0] = (long long)gridDim.x; tt_rec[1] = (long long)blockIdx.x;
            // kernel tag 0 = staged conv (stamps: start, first barrier, K loop done, epilogue done) | HW_ID << 8 | XCC_ID << 40
            tt_rec[2] = ((long long)(__builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4)) & 0xffffffffll) << 8 | (long long)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 0xf) << 40;
            tt_rec[3] = (long long)__builtin_amdgcn_s_memrealtime();
        }
    }
    TT_STAMP(0);
#endif
    const int in_len = uni(seg_len(a.in_seg, b));
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
    const rsrc_t wrs = make_rsrc(a.wb3, (unsigned)((size_t)(a.transposed ? a.out_stride : 1) * nsteps_all * nrt * ABLK));
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
        const bool v = col < W && pos >= 0 && pos < in_len;
        xoff[i] = v ? (unsigned)half * 8u * ld4 + (unsigned)pos * 4u : kOOB;
        lds_w[i] = sub * SUB + col * 32 + ((half ^ ((col >> 3) & 1)) << 4);
    }
    float xr[SPW][8];
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
                const rsrc_t rs = make_rsrc(xbase + (size_t)(c * NSUB + isub[i]) * CK * x_ld + in_base, (unsigned)((15ul * x_ld + in_len) * 4ul));
#pragma unroll
                for (int e = 0; e < 8; e++)
                    xr[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)xoff[i], (int)((unsigned)e * ld4), NTL ? 2 : 0));
            }
    };
    auto store_tile = [&](int bufi) {
        if ((STS_EXP & 8) && bufi >= 0 && as > 2) return;
        unsigned char* sb = smem3 + bufi * BUF;
#pragma unroll
        for (int i = 0; i < SPW; i++)
            if (sact[i]) {
                float v[8];
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
        return;
    }
    if ((STS_EXP & 32) && acc[0][0][0] != 12345.f) return;
    TT_STAMP(2);
    if (wvalid) tile_epilogue<MW, NW, NTL>(a, acc, mbase, n0 + wn * NW * 32, l31, half, n_count, out_len, out_base, phase, b);
#ifdef STS_TILE_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the stores have left
    TT_STAMP(3);
#endif
}

template <int MW, int NW, int WM, int WN, int NSUB, int KG = 1, int MATH = 0>
__global__ __launch_bounds__(WM* WN * KG * 64) __attribute__((amdgpu_waves_per_eu(MATH ? STS_H2_MINW : 1, MATH ? STS_H2_WAVES : 2))) void conv_bf3_kernel(ConvArgs a, int mtiles, int nx, int ny, int pm) {
    const TileId t = map_tile(nx, ny, a.B);
    if (!t.valid) return;
    conv_bf3_body<MW, NW, WM, WN, NSUB, KG, false, MATH>(a, mtiles, t.bx, t.by, t.bz, pm);
}

// grouped launch (layer d of all ResBlock chains of a stage in one grid), see conv_mfma_group_kernel
template <int MW, int NW, int WM, int WN, int NSUB, int KG = 1, int MATH = 0>
__global__ __launch_bounds__(WM* WN * KG * 64) __attribute__((amdgpu_waves_per_eu(MATH ? STS_H2_MINW : 1, MATH ? STS_H2_WAVES : 2))) void conv_bf3_group_kernel(ConvGroup G, int mtiles, int B, int nx, int ny, int interleave) {
    const TileId t = map_tile(nx, ny, B * G.n);
    if (!t.valid) return;
    // interleave: consecutive dispatch units belong to different members (different K lengths), so that workgroups that
    // share a CU do not run their load / MFMA / store phases in lockstep
    int gi, bx, b;
    if (interleave) { const int unit = t.bz * nx + t.bx; gi = unit % G.n; const int rest = unit / G.n; bx = rest % nx; b = rest / nx; }
    else { gi = t.bz / B; bx = t.bx; b = t.bz - gi * B; }
    const ConvArgs* ga = (const ConvArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    conv_bf3_body<MW, NW, WM, WN, NSUB, KG, false, MATH>(ga[gi], mtiles, bx, t.by, b);
}

// ------------------------------------------------------------------------------------------------
// fused ResBlock layer on the bf16 matrix cores (C = 32 MW WM <= 64)
//   y = x + conv2_{k2,d=1}( lrelu( conv1_{k1,d1}( lrelu(x) ) ) )        (ResBlock1.cpp:55-69, one dilation)
// With the matrix time cut to 6/16 these narrow stages are HBM-bound unless the intermediate stays on chip, and
// latency-bound unless a workgroup keeps many loads in flight.  So: the workgroup stages its WHOLE input window
// (all C channels x (P1 + 2 h1) positions), split, in one go -- every load of the tile is issued before the first
// is consumed, one barrier -- runs conv1 on P1 = 32 NW WN columns without another barrier, parks the biased,
// activated, zero-padded and split intermediate in LDS over the (dead) input window, and runs conv2 out of it.
// The intermediate is parked in the k order the accumulator layout gives for free (a lane holds rows 4 h + {0..3} and
// 8 + 4 h + {0..3} of every 16-row block = one 16-byte unit per plane); conv2's weights are packed to match (perm_k).
// ------------------------------------------------------------------------------------------------
template <int MW, int WM, int NW, int WN, int MATH = 0>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(2, 2))) void resblock_bf3_kernel(ResLayerGroup G, int nx, int wst, int interleave) {
    constexpr int C = 32 * MW * WM, NCH = C / 16, NRT = C / 32, NWAVE = WM * WN, P1 = 32 * NW * WN;
    constexpr int NPB = MATH ? 2 : 3;            // planes of a staged / parked activation
    constexpr int NPA = MATH ? 2 : 3;            // packed planes of a weight
    constexpr unsigned ABLK = NPA * 1024u;
    constexpr int PLANE2 = P1 * 32, CHUNK2 = NPB * PLANE2;
    constexpr int MAXSLOT = (P1 + MAX_HALO) / 32;
    constexpr int ITEMS = (NCH * MAXSLOT + NWAVE - 1) / NWAVE;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const TileId t = map_tile(nx, 1, G.B * G.n);
    if (!t.valid) return;
#ifdef STS_TILE_TRACE
    long long* tt_rec = nullptr;
    if (g_tile_trace && threadIdx.x == 0) {
        const unsigned slot = atomicAdd(&g_tile_trace_n, 1u);
        if (slot < g_tile_trace_cap) {
            tt_rec = g_tile_trace + (size_t)slot * 10;
            tt_rec[0] = (long long)gridDim.x; tt_rec[1] = (long long)blockIdx.x;
            tt_rec[2] = 1;        // fused layer (stamps: start, window staged, conv1 done, intermediate parked, conv2 done, epilogue done)
            tt_rec[3] = (long long)__builtin_amdgcn_s_memrealtime();
        }
    }
    TT_STAMP(0);
#endif
    int gi, tbx, b;
    if (interleave) { const int unit = t.bz * nx + t.bx; gi = unit % G.n; const int rest = unit / G.n; tbx = rest % nx; b = rest / nx; }
    else { gi = t.bz / G.B; tbx = t.bx; b = t.bz - gi * G.B; }
    const ResLayerArgs& a = ((const ResLayerArgs*)__builtin_amdgcn_kernarg_segment_ptr())[gi];
    const int d = a.dil1;
    const int h1 = d * (a.k1 - 1) / 2, h2 = (a.k2 - 1) / 2;
    const int NT = P1 - 2 * h2;
    const int len = seg_len(G.seg, b);
    const int n0 = tbx * NT;
    if (n0 >= len) return;
    const size_t base = (size_t)seg_start(G.seg, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const int wn = wave % WN, wm = wave / WN;
    const int mbase = wm * MW * 32;
    const int l31 = lane & 31, half = lane >> 5;
    const int W1 = P1 + 2 * h1, nslot = (W1 + 31) >> 5;
    const int w0 = n0 - h2 - h1;
    const int plane1 = wst * 32, chunk1 = NPB * plane1;
    const unsigned ld4 = (unsigned)G.ld * 4u;
    float amax = 0.f;

    // ---- stage the whole window: item t = (chunk, slot of 32 positions), wave w owns items w, w + NWAVE, ...
    {
        float xr[ITEMS][8];
        int lw[ITEMS];
        const int nitem = NCH * nslot;
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int it = swave + i * NWAVE;           // wave-uniform (scalar)
            const int c = it / nslot, sl = it - c * nslot;
            const int col = sl * 32 + l31, pos = w0 + col;
            const bool v = col < W1 && pos >= 0 && pos < len;
            const unsigned voff = v ? (unsigned)half * 8u * ld4 + (unsigned)pos * 4u : kOOB;
            lw[i] = c * chunk1 + col * 32 + ((half ^ ((col >> 3) & 1)) << 4);
            if (it < nitem) {
                const rsrc_t rs = make_rsrc(a.x + (size_t)c * CK * G.ld + base, (unsigned)((15ul * G.ld + len) * 4ul));
#pragma unroll
                for (int e = 0; e < 8; e++)
                    xr[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)((unsigned)e * ld4), 0));
            }
        }
#pragma unroll
        for (int i = 0; i < ITEMS; i++) {
            const int it = swave + i * NWAVE;
            if (it < nitem) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; e++) v[e] = xr[i][e] < 0.f ? xr[i][e] * G.slope : xr[i][e];
                if constexpr (MATH == 0) {
                    u32x4 ph, pm, pl;
                    split8(v, ph, pm, pl);
                    *(u32x4*)(smem3 + lw[i]) = ph;
                    *(u32x4*)(smem3 + plane1 + lw[i]) = pm;
                    *(u32x4*)(smem3 + 2 * plane1 + lw[i]) = pl;
                } else {
                    u32x4 ph, pl;
                    split8h(v, ph, pl, amax);
                    *(u32x4*)(smem3 + lw[i]) = ph;
                    *(u32x4*)(smem3 + plane1 + lw[i]) = pl;
                }
            }
        }
    }

    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; i++)
#pragma unroll
        for (int q = 0; q < NW; q++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][q][r] = 0.f;
    const unsigned a_voff = (unsigned)lane * 16u + (unsigned)(mbase >> 5) * ABLK;
    u32x4 fa[2][MW][NPA], fb[2][NW][NPB];
    auto mfmas = [&](u32x4 (&ac)[MW][NPA], u32x4 (&bc)[NW][NPB]) { step_mfmas<MATH, MW, NW, NPA, NPB>(acc, ac, bc); };
    __syncthreads();
    TT_STAMP(1);

    // ================= phase 1: conv1 on the P1 columns [n0 - h2, n0 - h2 + P1) =================
    {
        const int nsteps = NCH * a.k1;
        const rsrc_t wrs = make_rsrc(a.wb1, (unsigned)(nsteps * NRT) * ABLK);
        auto load_a = [&](int s, u32x4 (&dst)[MW][NPA]) {
            const unsigned sb = (unsigned)s * ((unsigned)NRT * ABLK);
#pragma unroll
            for (int i = 0; i < MW; i++)
#pragma unroll
                for (int pl = 0; pl < NPA; pl++)
                    dst[i][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)a_voff, (int)(sb + (unsigned)i * ABLK + (unsigned)(pl * 1024)), 0));
        };
        const int p0 = wn * NW * 32 + l31;
        auto load_b = [&](int c, int j, u32x4 (&dst)[NW][NPB]) {
            const int p = p0 + j * d;
            const unsigned char* sb = smem3 + c * chunk1 + p * 32 + ((half ^ ((p >> 3) & 1)) << 4);
#pragma unroll
            for (int q = 0; q < NW; q++)
#pragma unroll
                for (int pl = 0; pl < NPB; pl++) dst[q][pl] = *(const u32x4*)(sb + pl * plane1 + q * 1024);
        };
        int sj = 0, sc = 0;
        load_a(0, fa[0]);
        load_b(0, 0, fb[0]);
        for (int s = 0; s < nsteps; s += 2)
            static_for<0, 2>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if (s + u < nsteps) {
                    int nj = sj + 1, nc = sc;
                    if (nj == a.k1) { nj = 0; nc = sc + 1; }
                    load_a(s + u + 1, fa[(u + 1) % 2]);                       // past the end: zeros beyond the descriptor
                    load_b(nc < NCH ? nc : 0, nj, fb[(u + 1) % 2]);
                    __builtin_amdgcn_sched_barrier(0);      // keep the prefetches ahead of this step's MFMAs (the scheduler would sink them)
                    mfmas(fa[u % 2], fb[u % 2]);
                    sj = nj; sc = nc;
                }
            });
    }
    TT_STAMP(2);
    __syncthreads();          // every wave is done reading the staged window (the parked intermediate overwrites it)
    // ---- park: bias, conv2's input activation, conv2's zero padding outside [0, len), split
    static_for<0, MW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        float b1v[16];                                   // bias of the lane's 16 rows, requested together
#pragma unroll
        for (int r = 0; r < 16; r++) b1v[r] = 0.f;
        if (a.b1) {
#pragma unroll
            for (int r = 0; r < 16; r++) b1v[r] = a.b1[mbase + i * 32 + (r >> 3) * 16 + 8 * ((r & 7) >> 2) + 4 * half + (r & 3)];
        }
        static_for<0, NW>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const int col = wn * NW * 32 + q * 32 + l31;
            const int pos = n0 - h2 + col;
            const bool inside = pos >= 0 && pos < len;
            static_for<0, 2>([&](auto hc) {
                constexpr int hh = decltype(hc)::value;       // 16-row block of the 32-row tile
                const int cc = (mbase >> 4) + 2 * i + hh;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    float t1 = (MATH ? acc[i][q][hh * 8 + e] * a.ws1 : acc[i][q][hh * 8 + e]) + b1v[hh * 8 + e];
                    t1 = t1 < 0.f ? t1 * G.slope : t1;
                    v[e] = inside ? t1 : 0.f;
                    acc[i][q][hh * 8 + e] = 0.f;
                }
                unsigned char* dst = smem3 + cc * CHUNK2 + col * 32 + ((half ^ ((col >> 3) & 1)) << 4);
                if constexpr (MATH == 0) {
                    u32x4 ph, pm, pl;
                    split8(v, ph, pm, pl);
                    *(u32x4*)(dst) = ph;
                    *(u32x4*)(dst + PLANE2) = pm;
                    *(u32x4*)(dst + 2 * PLANE2) = pl;
                } else {
                    u32x4 ph, pl;
                    split8h(v, ph, pl, amax);
                    *(u32x4*)(dst) = ph;
                    *(u32x4*)(dst + PLANE2) = pl;
                }
            });
        });
    });
    __syncthreads();
    TT_STAMP(3);

    // The residual (raw x of the output columns; the staged copy was activated and split) and conv2's bias are requested BEFORE
    // conv2's K loop, in the transposed-quad layout of the epilogue below: they arrive under the MFMAs instead of costing the
    // epilogue a memory round trip (round 3, tools/tile_trace.py: epilogue 4.4 us of a 19 us 32-channel tile)
    const int lane4 = l31 & 3, m4 = (l31 >> 2) * 4;
    f32x4u xres[MW][NW][4];
    float b2v[MW][4];
#pragma unroll
    for (int i = 0; i < MW; i++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const int row = mbase + i * 32 + 8 * g + 4 * half + lane4;
            b2v[i][g] = a.b2 ? a.b2[row] : 0.f;
#pragma unroll
            for (int q = 0; q < NW; q++) {
                const int col = wn * NW * 32 + q * 32 + m4;
                const int pos = n0 + col;
                xres[i][q][g] = f32x4u{0.f, 0.f, 0.f, 0.f};
                if (col < NT && pos < len) {
                    const float* xp = a.x + (size_t)row * G.ld + base + pos;
                    if (col + 3 < NT && pos + 3 < len) xres[i][q][g] = *(const f32x4u*)xp;
                    else { for (int e = 0; e < 4; e++) if (col + e < NT && pos + e < len) xres[i][q][g][e] = xp[e]; }
                }
            }
        }

    // ================= phase 2: conv2 (dilation 1) out of the parked intermediate =================
    {
        const int nsteps = NCH * a.k2;
        const rsrc_t wrs = make_rsrc(a.wb2, (unsigned)(nsteps * NRT) * ABLK);
        auto load_a = [&](int s, u32x4 (&dst)[MW][NPA]) {
            const unsigned sb = (unsigned)s * ((unsigned)NRT * ABLK);
#pragma unroll
            for (int i = 0; i < MW; i++)
#pragma unroll
                for (int pl = 0; pl < NPA; pl++)
                    dst[i][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)a_voff, (int)(sb + (unsigned)i * ABLK + (unsigned)(pl * 1024)), 0));
        };
        const int p0 = wn * NW * 32 + l31;
        auto load_b = [&](int c, int j, u32x4 (&dst)[NW][NPB]) {
            const int p = p0 + j;
            const unsigned char* sb = smem3 + c * CHUNK2 + p * 32 + ((half ^ ((p >> 3) & 1)) << 4);
#pragma unroll
            for (int q = 0; q < NW; q++)
#pragma unroll
                for (int pl = 0; pl < NPB; pl++) dst[q][pl] = *(const u32x4*)(sb + pl * PLANE2 + q * 1024);
        };
        int sj = 0, sc = 0;
        load_a(0, fa[0]);
        load_b(0, 0, fb[0]);
        for (int s = 0; s < nsteps; s += 2)
            static_for<0, 2>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if (s + u < nsteps) {
                    int nj = sj + 1, nc = sc;
                    if (nj == a.k2) { nj = 0; nc = sc + 1; }
                    load_a(s + u + 1, fa[(u + 1) % 2]);
                    load_b(nc < NCH ? nc : 0, nj, fb[(u + 1) % 2]);
                    __builtin_amdgcn_sched_barrier(0);      // keep the prefetches ahead of this step's MFMAs (the scheduler would sink them)
                    mfmas(fa[u % 2], fb[u % 2]);
                    sj = nj; sc = nc;
                }
            });
    }
    TT_STAMP(4);
    // ---- epilogue: + b2 + x (the residual is re-read: the staged copy was activated and split).  As in tile_epilogue
    // (conv_common.hpp): a 4 x 4 transpose inside the lane quads turns a lane's 4 consecutive rows of one column into 4 consecutive
    // columns of one row, so the tile's residual arrives and its result leaves through 16-byte accesses (a quarter of the memory
    // instructions; the epilogue was store-issue-bound)
    {
        static_for<0, MW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            static_for<0, NW>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const int col = wn * NW * 32 + q * 32 + m4;
                const int pos = n0 + col;
                const bool any = col < NT && pos < len, full = col + 3 < NT && pos + 3 < len;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    float w[4] = {acc[i][q][4 * g], acc[i][q][4 * g + 1], acc[i][q][4 * g + 2], acc[i][q][4 * g + 3]};
                    quad_transpose(w, l31);
                    if (any) {
                        f32x4u o;
#pragma unroll
                        for (int e = 0; e < 4; e++) o[e] = (MATH ? w[e] * a.ws2 : w[e]) + b2v[i][g] + xres[i][q][g][e];
                        float* yp = a.y + (size_t)(mbase + i * 32 + 8 * g + 4 * half + lane4) * G.ld + base + pos;
                        if (full) *(f32x4u*)yp = o;
                        else { for (int e = 0; e < 4; e++) if (col + e < NT && pos + e < len) yp[e] = o[e]; }
                    }
                }
            });
        });
    }
    if constexpr (MATH == 1) { if (amax > kH2Limit && G.ovf) *G.ovf = 1u; }
#ifdef STS_TILE_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TT_STAMP(5);
#endif
}

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
    if (g_tile_trace && tid == 0) {
        const unsigned slot = atomicAdd(&g_tile_trace_n, 1u);
        if (slot < g_tile_trace_cap) {
            long long* r = g_tile_trace + (size_t)slot * 10;
            r[0] = (long long)gridDim.x; r[1] = (long long)blockIdx.x; r[2] = 2 | ((long long)x << 40); r[3] = (long long)__builtin_amdgcn_s_memrealtime();
            r[4] = ps_t0; r[5] = (long long)__builtin_amdgcn_s_memtime(); r[6] = ps_wait; r[7] = ps_items; r[8] = 0; r[9] = 0;
        }
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

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static inline void split_host(float x, uint16_t (&p)[3]) {
    uint32_t u; memcpy(&u, &x, 4);
    const uint32_t h = u & 0xffff0000u;
    float hf; memcpy(&hf, &h, 4);
    const float r = x - hf;
    uint32_t v; memcpy(&v, &r, 4);
    const uint32_t m = v & 0xffff0000u;
    float mf; memcpy(&mf, &m, 4);
    const float l = r - mf;
    uint32_t w; memcpy(&w, &l, 4);
    p[0] = (uint16_t)(h >> 16); p[1] = (uint16_t)(m >> 16); p[2] = (uint16_t)(w >> 16);
}

// fp16 terms of a scaled weight (MATH 1, see split8h): P0 = fp16(ws), P1 = fp16(ws - P0), P2 = P0 * 2^-11
static inline uint16_t f16_bits(float x) { const _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; }
static inline void split_host_h2(float ws, uint16_t (&p)[3]) {
    const _Float16 h = (_Float16)ws;
    p[0] = f16_bits(ws);
    p[1] = f16_bits(ws - (float)h);
    p[2] = 0;                            // (P2 = P0 * 2^-11 is formed in the kernel)
}

size_t bf3_pack(const float* wp, int nphase, int ntap, int Cin_pad, int Cout_pad, void* dst, bool perm_k, int math, float* wscale) {
    const int nchunk = Cin_pad / CK, nrt = Cout_pad / 32;
    const int npl = math == 1 ? 2 : 3;                     // packed planes
    const size_t bytes = (size_t)nphase * nchunk * ntap * nrt * npl * 1024;
    if (!dst) return bytes;
    uint16_t* d = (uint16_t*)dst;
    float up = 1.0f;
    if (math == 1) {
        // per-conv power of two that brings the largest weight into [2^13, 2^14): the small terms of all but vanishing weights
        // stay fp16-normal, the tile is scaled back by 1 / up (exact) in the kernel
        float mx = 0.f;
        const size_t n = (size_t)nphase * ntap * Cin_pad * Cout_pad;
        for (size_t i = 0; i < n; i++) { const float v = wp[i] < 0.f ? -wp[i] : wp[i]; if (v > mx) mx = v; }
        int e = 0;
        if (mx > 0.f && mx < 3.0e38f) {                         // mx = f * 2^e, f in [0.5, 1)
            (void)frexpf(mx, &e);
            const int s = 14 - e;
            up = ldexpf(1.0f, s > 100 ? 100 : (s < -100 ? -100 : s));    // (a conv of vanishing weights: keep the scale and its inverse finite)
        }
        if (wscale) *wscale = 1.0f / up;
    }
    for (int ph = 0; ph < nphase; ph++)
        for (int c = 0; c < nchunk; c++)
            for (int j = 0; j < ntap; j++)
                for (int rt = 0; rt < nrt; rt++) {
                    uint16_t* blk = d + ((((size_t)ph * nchunk + c) * ntap + j) * nrt + rt) * (size_t)(npl * 512);   // planes x 512 values
                    const float* slab = wp + ((size_t)ph * ntap + j) * Cin_pad * Cout_pad;
                    for (int l = 0; l < 64; l++) {
                        const int i = l & 31, h = l >> 5;
                        for (int e = 0; e < 8; e++) {
                            uint16_t p[3];
                            const int ci = perm_k ? 8 * (e >> 2) + 4 * h + (e & 3) : 8 * h + e;
                            const float wv = slab[(size_t)(c * CK + ci) * Cout_pad + rt * 32 + i];
                            if (math == 1) split_host_h2(wv * up, p); else split_host(wv, p);
                            for (int pl = 0; pl < npl; pl++) blk[pl * 512 + l * 8 + e] = p[pl];
                        }
                    }
                }
    return bytes;
}

// tile codes: t = 0..5 below with 16-channel chunks, 8 + t the same tile with 32-channel chunks (NSUB = 2)
//   0: 128 x 128 (4 waves of 64 x 64)   1: 64 x 256 (4 waves)   2: 128 x 256 (8 waves)   3: 64 x 128 (2 waves)
//   4:  32 x 256 (4 waves of 32 x 64)   5: 32 x 128 (2 waves)
constexpr int kNumBf3Tiles = 8;      // 6 / 7 (round 3): a wave owns 32 rows x 128 columns -- no two waves of a workgroup fetch the same weight rows
static bool h2_tile(int tile) { return tile == 0 || tile == 3 || tile == 4 || tile == 20 || tile == 22 || tile == 23 || tile == 24; }   // built for MATH 1
static bool bf3_tile_ok(int tile) { return (tile >= 0 && tile < kNumBf3Tiles) || (tile >= 8 && tile < 8 + kNumBf3Tiles) || (tile >= 20 && tile < 25); }

bool conv_bf3_eligible(const ConvArgs& a) {
    if (!a.wb3 || a.depthwise || a.in_reflect) return false;
    if (a.Cin != a.Cin_pad || a.Cin_pad % CK != 0 || a.Cout_pad % 32 != 0 || a.Cin < 32) return false;
    if ((double)a.x_ld * 64.0 >= 4.0e9) return false;       // 16 rows of a chunk behind one 32-bit buffer descriptor
    const int first = a.tap_off, last = a.tap_off + (a.ntap - 1) * a.tap_step;
    const int halo = first < last ? last - first : first - last;
    if (halo > MAX_HALO) return false;
    if (a.epi == EPI_GATE && !a.gate_perm) return false;
    if (a.epi == EPI_TANH_PCM || a.kslices > 1) return false;
    return true;
}

// Tile choice (tools/bench_variants.sh sweeps on MI355X): all rows of a <= 128-row block in one workgroup
// (the staged window is split once) when that still yields >= 2 workgroups per CU; a grid-starved launch (the
// 256-channel stage of one utterance: 252 such tiles) takes 32-row x 256-column tiles instead (4x the workgroups,
// 3 waves per SIMD).
static int pick_bf3_tile(int Cout_pad, long max_n, long units, bool transposed = false, int math = 0) {
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

// H2: this tile is also built for the two-term fp16 arithmetic (the tiles the automatic choice uses plus 24; conv_bf3 /
// conv_bf3_group send a MATH 1 conv to no other.  Round 3's sweep of the rest under MATH 1 -- 32 x 128 per wave, 32-channel
// chunks, 64-row tiles -- found nothing better: profiles/r03_f16x2_tile_sweep.log)
template <int MW, int NW, int WM, int WN, int NSUB, int KG = 1, bool H2 = false>
static void launch_bf3(const ConvArgs& a, int nphase, hipStream_t st, int pm = 0) {
    constexpr int MT = 32 * MW * WM, NT = 32 * NW * WN;
    pm = pm && a.transposed && (MW == 1 || a.Cout_pad % (32 * MW) == 0);    // a wave's rows must lie inside one phase
    const int mt = pm ? (a.Cout_pad * nphase + MT - 1) / MT : (a.Cout_pad + MT - 1) / MT;
    const int nx = (a.max_n + NT - 1) / NT, ny = pm ? mt : mt * nphase;
    if constexpr (H2) {
        if (a.math == 1) {
            const size_t lds = bf3_lds_bytes<MW, NW, WM, WN, NSUB, KG, 1>();
            hipLaunchKernelGGL((conv_bf3_kernel<MW, NW, WM, WN, NSUB, KG, 1>), dim3(mapped_grid(nx, ny, a.B)), dim3(WM * WN * KG * 64), lds, st, a, mt, nx, ny, pm);
            return;
        }
    }
    const size_t lds = bf3_lds_bytes<MW, NW, WM, WN, NSUB, KG>();
    hipLaunchKernelGGL((conv_bf3_kernel<MW, NW, WM, WN, NSUB, KG>), dim3(mapped_grid(nx, ny, a.B)), dim3(WM * WN * KG * 64), lds, st, a, mt, nx, ny, pm);
}
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

// 20: 128 x 128 with K split over two wave groups inside the workgroup (8 waves, 32-channel staged chunks)
// 24: 256 x 64, K split over two wave groups (8 waves): all rows of a 256-channel conv behind ONE staged window
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

long conv_bf3_blocks(const ConvArgs& a) {
    const int nphase = a.transposed ? a.out_stride : 1;
    const int tile = pick_bf3_tile(a.Cout_pad, a.max_n, (long)a.B * nphase, a.transposed != 0, a.math);
    const int mt = tile == 4 ? 32 : (tile == 3 ? 64 : 128), nt = tile == 4 ? 256 : 128;
    return (long)((a.max_n + nt - 1) / nt) * ((a.Cout_pad + mt - 1) / mt) * nphase * a.B;
}

void conv_bf3(const ConvArgs& a, hipStream_t st, int tile) {
    const int nphase = a.transposed ? a.out_stride : 1;
    if (a.max_n <= 0 || a.B <= 0) return;
    if (!bf3_tile_ok(tile) || (a.math == 1 && !h2_tile(tile))) tile = pick_bf3_tile(a.Cout_pad, a.max_n, (long)a.B * nphase, a.transposed != 0, a.math);
    if (tile >= 8 && tile < 16 && a.Cin_pad % 32 != 0) tile -= 8;
    switch (tile) {
        case 20: if (a.Cin_pad % 32 == 0) { launch_bf3<2, 2, 2, 2, 2, 2, true>(a, nphase, st); break; } launch_bf3<2, 2, 2, 2, 1, 1, true>(a, nphase, st); break;
        // phase-merged rows (transposed convs): 21: 256 x 128 (8 waves)   22: 128 x 128   23: 64 x 128 as two 32-row waves x 2
        case 24: if (a.Cin_pad % 32 == 0) { launch_bf3<2, 2, 4, 1, 2, 2, true>(a, nphase, st); break; } launch_bf3<2, 2, 2, 2, 1, 1, true>(a, nphase, st); break;
        case 21: launch_bf3<2, 2, 4, 2, 1>(a, nphase, st, 1); break;
        case 22: launch_bf3<2, 2, 2, 2, 1, 1, true>(a, nphase, st, 1); break;
        case 23: launch_bf3<1, 2, 2, 2, 1, 1, true>(a, nphase, st, 1); break;
        case 0: launch_bf3<2, 2, 2, 2, 1, 1, true>(a, nphase, st); break;
        case 1: launch_bf3<2, 2, 1, 4, 1>(a, nphase, st); break;
        case 2: launch_bf3<2, 2, 2, 4, 1>(a, nphase, st); break;
        case 3: launch_bf3<2, 2, 1, 2, 1, 1, true>(a, nphase, st); break;
        case 4: launch_bf3<1, 2, 1, 4, 1, 1, true>(a, nphase, st); break;
        case 5: launch_bf3<1, 2, 1, 2, 1>(a, nphase, st); break;
        case 6: launch_bf3<1, 4, 4, 1, 1>(a, nphase, st); break;      // 128 x 128, four waves of 32 x 128
        case 7: launch_bf3<1, 4, 2, 1, 1>(a, nphase, st); break;      //  64 x 128, two waves of 32 x 128
        case 14: launch_bf3<1, 4, 4, 1, 2>(a, nphase, st); break;
        case 15: launch_bf3<1, 4, 2, 1, 2>(a, nphase, st); break;
        case 8: launch_bf3<2, 2, 2, 2, 2>(a, nphase, st); break;
        case 9: launch_bf3<2, 2, 1, 4, 2>(a, nphase, st); break;
        case 10: launch_bf3<2, 2, 2, 4, 2>(a, nphase, st); break;
        case 11: launch_bf3<2, 2, 1, 2, 2>(a, nphase, st); break;
        case 12: launch_bf3<1, 2, 1, 4, 2>(a, nphase, st); break;
        default: launch_bf3<1, 2, 1, 2, 2>(a, nphase, st); break;
    }
}

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
    if (!bf3_tile_ok(tile) || (a.math == 1 && !h2_tile(tile))) tile = pick_bf3_tile(a.Cout_pad, a.max_n, (long)a.B * G.n, false, a.math);
    if (tile >= 8 && tile < 16) for (int i = 0; i < G.n; i++) if (G.g[i].Cin_pad % 32 != 0) { tile -= 8; break; }
    switch (tile) {
        case 20: { bool ok = true; for (int i = 0; i < G.n; i++) ok = ok && G.g[i].Cin_pad % 32 == 0;
                   if (ok) launch_bf3_group<2, 2, 2, 2, 2, 2, true>(G, st); else launch_bf3_group<2, 2, 2, 2, 1, 1, true>(G, st); break; }
        case 24: { bool ok = true; for (int i = 0; i < G.n; i++) ok = ok && G.g[i].Cin_pad % 32 == 0;
                   if (ok) launch_bf3_group<2, 2, 4, 1, 2, 2, true>(G, st); else launch_bf3_group<2, 2, 2, 2, 1, 1, true>(G, st); break; }
        case 0: launch_bf3_group<2, 2, 2, 2, 1, 1, true>(G, st); break;
        case 1: launch_bf3_group<2, 2, 1, 4, 1>(G, st); break;
        case 2: launch_bf3_group<2, 2, 2, 4, 1>(G, st); break;
        case 3: launch_bf3_group<2, 2, 1, 2, 1, 1, true>(G, st); break;
        case 4: launch_bf3_group<1, 2, 1, 4, 1, 1, true>(G, st); break;
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
        default: launch_bf3_group<1, 2, 1, 2, 2>(G, st); break;
    }
}

bool resblock_bf3_eligible(const ResLayerGroup& G) {
    if (G.n < 1 || G.n > kMaxGroup || (G.C != 32 && G.C != 64 && G.C != 128) || G.max_n <= 0 || G.B <= 0) return false;
    if ((double)G.ld * 64.0 >= 4.0e9) return false;
    for (int i = 0; i < G.n; i++) {
        const ResLayerArgs& a = G.g[i];
        if (!a.wb1 || !a.wb2 || !(a.k1 & 1) || !(a.k2 & 1) || a.k1 < 1 || a.k2 < 1) return false;
        if (a.dil1 < 1 || a.dil1 * (a.k1 - 1) > MAX_HALO || a.k2 - 1 > 32) return false;
        if (a.x == a.y) return false;
    }
    return true;
}

template <int MW, int WM, int NW, int WN>
static void launch_resblock_bf3(const ResLayerGroup& G, hipStream_t st) {
    constexpr int C = 32 * MW * WM, P1 = 32 * NW * WN;
    int nx = 0, halo = 0;
    for (int i = 0; i < G.n; i++) {
        const int NT = P1 - (G.g[i].k2 - 1);
        const int n = (G.max_n + NT - 1) / NT;
        if (n > nx) nx = n;
        const int h = G.g[i].dil1 * (G.g[i].k1 - 1);
        if (h > halo) halo = h;
    }
    const int wst = (P1 + halo + 31) / 32 * 32;
    const size_t pb = G.math == 1 ? 4 : 6;                                          // bytes per staged value: its fp16 / bf16 terms
    const size_t stage = (size_t)C * wst * pb, park = (size_t)C * P1 * pb + 1024;   // + slack: conv2's taps of the discarded last columns
    const size_t lds = stage > park ? stage : park;
    static const int il = exp_int("STS_BF3_INTERLEAVE", 0);
    if (G.math == 1)
        hipLaunchKernelGGL((resblock_bf3_kernel<MW, WM, NW, WN, 1>), dim3(mapped_grid(nx, 1, G.B * G.n)), dim3(64 * WM * WN), lds, st, G, nx, wst, (il >> 1) & 1);
    else
        hipLaunchKernelGGL((resblock_bf3_kernel<MW, WM, NW, WN>), dim3(mapped_grid(nx, 1, G.B * G.n)), dim3(64 * WM * WN), lds, st, G, nx, wst, (il >> 1) & 1);
}

// variant: -1 automatic; C = 64: 0 = (32 x 64 per wave, 2 x 2 waves), 1 = (64 x 64 per wave, 1 x 2 waves);
//          C = 32: 0 = 4 waves x 64 columns (P1 = 256), 1 = 2 waves x 64 columns (P1 = 128)
void resblock_bf3(const ResLayerGroup& Gin, hipStream_t st, int variant) {
    ResLayerGroup G = Gin;
    for (int i = 1; i < G.n; i++)                       // longest K loops first
        for (int j = i; j > 0 && G.g[j].k1 + G.g[j].k2 > G.g[j - 1].k1 + G.g[j - 1].k2; j--) {
            ResLayerArgs t = G.g[j]; G.g[j] = G.g[j - 1]; G.g[j - 1] = t;
        }
    if (G.C == 128) {
        // the whole 128-channel window (147 KB of the CU's 160 KB LDS): one 8-wave workgroup per CU
        if (variant == 1) launch_resblock_bf3<2, 2, 2, 2>(G, st); else launch_resblock_bf3<1, 4, 2, 2>(G, st);
    } else if (G.C == 64) {
        if (variant == 1) launch_resblock_bf3<2, 1, 2, 2>(G, st); else launch_resblock_bf3<1, 2, 2, 2>(G, st);
    } else {
        if (variant == 1) launch_resblock_bf3<1, 1, 2, 2>(G, st); else launch_resblock_bf3<1, 1, 2, 4>(G, st);
    }
}

}  // namespace sts
