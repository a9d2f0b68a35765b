// conv.hip -- Conv1d / ConvTranspose1d for gfx950 (CDNA4, wave64).
//
// Replaces /root/reference/src/nn_op/nn_conv1d.cpp:118-199 (zero-stuffed im2col + Eigen GEMM) and
// nn_conv1d_transposed.cpp:106-150 (GEMM + scatter-add), which carry ~95 % of the reference's
// run time.  MI355X-first design instead of a translation:
//   * implicit GEMM on the matrix cores with the exact-fp32 instruction v_mfma_f32_32x32x2_f32:
//     M = output channels, N = time, K = (tap, input channel).  Only the TRUE taps are iterated
//     (the reference multiplies by the zero-stuffed dilated kernel).
//   * the input tile plus its dilated halo is staged ONCE per 16-channel chunk in a double-buffered LDS
//     tile (raw-buffer loads: the hardware range check is the zero padding; fused input leaky-relu);
//     the B operand of every tap is the same LDS rows read at a shifted column (bank-conflict free:
//     32 consecutive floats per half-wave).  One barrier per chunk.
//   * weights are repacked at load time to [tap][cin][cout] so the A operand is a coalesced 128-B row
//     read straight from L2 (one float per lane per MFMA) through a buffer descriptor with the
//     (tap, channel) part of the address in the scalar offset; A and B fragments ping-pong between two
//     register sets so step s+1's operands are in flight under step s's 16 MFMAs.
//   * bias, per-utterance conditioning, residual add, the WaveNet gate tanh*sigmoid, res/skip split
//     and the flow's "x1 -= m" are epilogues on the accumulator registers -- no elementwise kernels,
//     no extra HBM round trips.
//   * ConvTranspose1d is polyphase: phase p of the output only touches taps k == p (mod stride),
//     so it is the same kernel with tap_step = -1 and an output stride.
//   * launches that would not fill the chip (text encoder, flow, duration predictor at batch 1) go to
//     a split-K sibling (conv_mfma_splitk_kernel): one output tile per workgroup, K split over up to
//     16 waves, operands streamed from L2 through a register ring, no LDS staging.
// A plain VALU kernel (conv_generic) covers what the matrix cores cannot fill (Cin < 32, depthwise) and
// conv_cout1_kernel the long single-output-channel FIR at the end of HiFi-GAN (fused tanh + int16).
#include "kernels.hpp"
#include "devmath.hpp"
#include "conv_common.hpp"

namespace sts {

// logical input sample (zero padding, optional reflect-left-1 view, fused input activation)
__device__ __forceinline__ float load_in(const ConvArgs& a, const float* xrow, int pos, int in_len, int orig_len) {
    if (pos < 0 || pos >= in_len) return 0.f;
    int src = pos;
    if (a.in_reflect) {
        src = pos - 1;
        if (src < 0) { if (orig_len > 1) src = 1; else return 0.f; }
    }
    float v = xrow[src];
    if (a.in_act) v = v < 0.f ? v * a.in_slope : v;
    return v;
}

// ------------------------------------------------------------------------------------------------
// matrix-core kernel
// ------------------------------------------------------------------------------------------------
#ifndef STS_EXP
#define STS_EXP 0   // timing experiments only (tools/exp_build.sh); 0 in every shipped build
#endif

#ifndef STS_RA_GROUP
#define STS_RA_GROUP 3   // A-fragment ring depth of the grouped (ResBlock) launches; experiment knob of tools/exp_build.sh
#endif
template <int MW, int NW, int WM, int WN, int RA = 3>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs& a, const int mtiles, const int bx, const int by, const int b) {
    constexpr int MT = 32 * MW * WM, NT = 32 * NW * WN, NTHR = WM * WN * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int orig_len = uni(seg_len(a.in_seg, b));
    const int in_len = orig_len + (a.in_reflect ? 1 : 0);
    const int out_len = uni(seg_len(a.out_seg, b));
    const int n_count = a.transposed ? in_len + a.n_extra : out_len;
    const int n0 = bx * NT;
    if (n0 >= n_count) return;
    const int phase = by / mtiles;
    const int m0 = (by - phase * mtiles) * MT;
    const size_t in_base = (size_t)uni(seg_start(a.in_seg, b)), out_base = (size_t)uni(seg_start(a.out_seg, b));
    const float* const xbase = uni(a.x);
    const long x_ld = uni(a.x_ld);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;

    const int first = a.tap_off, last = a.tap_off + (a.ntap - 1) * a.tap_step;
    const int lo = first < last ? first : last, hi = first < last ? last : first;
    const int W = NT + (hi - lo);      // staged window width (<= NT + MAX_HALO)
    constexpr int WIN = NT + MAX_HALO;   // staged columns per row; thread t owns columns t, t + NTHR, ... < WIN
    constexpr int ldsw = WIN;
    const int win0 = n0 + lo;

    const float* w = a.w + (size_t)phase * a.ntap * a.Cin_pad * a.Cout_pad;
    const int mbase = m0 + wm * MW * 32;
    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; i++)
#pragma unroll
        for (int j = 0; j < NW; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nchunk = a.Cin_pad / CK;
    const int nsteps = nchunk * a.ntap;

    // A fragment of step s = chunk*ntap + tap : lane holds W[tap][chunk*CK + 2p + half][mbase + mw*32 + l31].
    // A fragment through a raw buffer descriptor: one per-lane VGPR offset (row parity + column) for the
    // whole kernel, the (tap, channel pair, row tile) part of the address rides in the scalar offset --
    // no 64-bit vector address arithmetic per load.  Row tiles beyond Cout_pad fall outside the
    // descriptor and read as 0.
    const rsrc_t wrs = make_rsrc(w, (unsigned)((size_t)a.ntap * a.Cin_pad * a.Cout_pad * 4));
    const unsigned a_voff = (unsigned)(((size_t)half * a.Cout_pad + mbase + l31) * 4);
    auto load_a = [&](int c, int j, float (&dst)[CK / 2][MW]) {
        const unsigned sbase = (unsigned)((((size_t)j * a.Cin_pad + (size_t)c * CK) * a.Cout_pad) * 4);
#pragma unroll
        for (int p = 0; p < CK / 2; p++)
#pragma unroll
            for (int i = 0; i < MW; i++)
                dst[p][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    wrs, (int)a_voff, (int)(sbase + (unsigned)(2 * p) * (unsigned)a.Cout_pad * 4u + (unsigned)i * 128u), 0));
    };
    // B fragment of tap j: the staged LDS rows read at the tap's column shift
    // the LDS tile is double-buffered (chunk c lives in buffer c & 1): the next chunk's tile is written
    // while the current chunk's last MFMAs issue, and a single barrier per chunk publishes it
    auto load_b = [&](int bufi, int j, float (&dst)[CK / 2][NW]) {
        const float* sb = smem + bufi * (CK * ldsw);
        const int bcol0 = wn * NW * 32 + l31 + j * a.tap_step + a.tap_off - lo;
#pragma unroll
        for (int p = 0; p < CK / 2; p++)
#pragma unroll
            for (int q = 0; q < NW; q++) dst[p][q] = sb[(2 * p + half) * ldsw + bcol0 + q * 32];
    };

    // ---- input staging: every thread owns RI columns of the window for all CK rows of a chunk.
    // The (column -> source index, validity) map does not depend on the chunk, so it is computed
    // once; the loads of a whole chunk are issued back to back into registers (no load->store
    // serialisation) and the NEXT chunk's loads are in flight while the current chunk's MFMAs issue.
    constexpr int RI = (NT + MAX_HALO + NTHR - 1) / NTHR;
    unsigned xoff[RI];
#pragma unroll
    for (int i = 0; i < RI; i++) {
        const int col = tid + i * NTHR;
        const int pos = win0 + col;
        bool v = col < W && pos >= 0 && pos < in_len;
        int src = pos;
        if (a.in_reflect) { src = pos - 1; if (src < 0) { src = 1; v = v && orig_len > 1; } }
        xoff[i] = v ? (unsigned)src * 4u : kOOB;
    }
    float xr[CK][RI];
    int sj = 0, sc = 0;   // tap / chunk of the current step
    auto load_x = [&](int c) {
        if (STS_EXP & 1) { if (c > 0) return; }
#pragma unroll
        for (int i = 0; i < RI; i++)
            if ((i + 1) * NTHR <= WIN || tid < WIN - i * NTHR) {   // partial last round: whole waves drop out
#pragma unroll
                for (int r = 0; r < CK; r++) {
                    const int ci = c * CK + r;
                    const rsrc_t rs = make_rsrc(xbase + (size_t)ci * x_ld + in_base, ci < a.Cin ? (unsigned)orig_len * 4u : 0u);
                    xr[r][i] = buf_load(rs, xoff[i]);   // raw: activation is applied at store time
                }
            }
    };
    auto store_tile = [&](int bufi) {      // registers (chunk loaded earlier) -> LDS buffer, fused input activation
        if (STS_EXP & 8) { if (bufi >= 0 && sc > 0) return; }
        float* sb = smem + bufi * (CK * ldsw);
#pragma unroll
        for (int i = 0; i < RI; i++)
            if ((i + 1) * NTHR <= WIN || tid < WIN - i * NTHR) {
#pragma unroll
                for (int r = 0; r < CK; r++) {
                    float v = xr[r][i];
                    if (a.in_act) v = v < 0.f ? v * a.in_slope : v;
                    sb[r * ldsw + tid + i * NTHR] = v;
                }
            }
    };

    // ---- main loop over steps (chunk, tap).  Fragment registers form rings indexed at compile time
    // (the loop is unrolled by 6 = lcm(3, 2)): the A fragment (L2) of step s+2 and the B fragment (LDS) of
    // step s+1 are in flight while step s issues its MFMAs.  The A ring is three deep for the sake of the
    // INPUT prefetch: vmcnt retires in order, so a wait for an A fragment issued after the next chunk's
    // input loads also waits for those (HBM latency); with distance 2 the first such wait comes three
    // steps after the input loads were issued instead of one.
    // (RA = ring depth of the A fragments, prefetch distance RA - 1 steps; the loop is unrolled by UNR = lcm(RA, 2))
    constexpr int UNR = (RA % 2 == 0) ? RA : 2 * RA;
    float fa[RA][CK / 2][MW], fb[2][CK / 2][NW];
    int aj = 0, ac = 0;   // tap / chunk of the next A fragment to request
    auto request_a = [&](float (&dst)[CK / 2][MW]) {
        if (!(STS_EXP & 2) || (ac == 0 && aj < 2)) load_a(ac, aj, dst);
        if (++aj == a.ntap) { aj = 0; ac++; }
    };
    auto do_step = [&](float (&acur)[CK / 2][MW], float (&anew)[CK / 2][MW], float (&bcur)[CK / 2][NW],
                       float (&bnxt)[CK / 2][NW], int s) {
        const bool last_tap = sj + 1 == a.ntap;
        int nj = sj + 1, nc = sc;
        if (last_tap) { nj = 0; nc = sc + 1; }
        // the prefetches are unconditional: past the last step they read inside the weight descriptor / the LDS
        // tile and are never used (predicated loads would cost register merges in the steady state)
        request_a(anew);
        {
            if (last_tap && s + 1 < nsteps) {
                // chunk boundary: this step's B fragment is already in registers, so the next tile can be
                // published before this step's MFMAs issue
                store_tile(nc & 1);          // chunk nc's tile (in registers since the start of chunk sc)
                if (!(STS_EXP & 4)) __syncthreads();   // tile nc visible to all waves; everyone is done reading tile sc
                if (nc + 1 < nchunk) load_x(nc + 1);
            }
            if (!(STS_EXP & 16)) load_b(nc & 1, nj, bnxt);
        }
#pragma unroll
        for (int p = 0; p < CK / 2; p++)
#pragma unroll
            for (int i = 0; i < MW; i++)
#pragma unroll
                for (int q = 0; q < NW; q++)
                    acc[i][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[p][i], bcur[p][q], acc[i][q], 0, 0, 0);
        sj = nj; sc = nc;
    };
    load_x(0);
    static_for<0, RA - 1>([&](auto rc) { constexpr int r = decltype(rc)::value; if (r == 0 || nsteps > r) request_a(fa[r]); });
    store_tile(0);
    __syncthreads();
    load_b(0, 0, fb[0]);
    if (nchunk > 1) load_x(1);
    for (int s = 0; s < nsteps; s += UNR)
        static_for<0, UNR>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if (s + u < nsteps) do_step(fa[u % RA], fa[(u + RA - 1) % RA], fb[u % 2], fb[(u + 1) % 2], s + u);
        });

    // ---- epilogue on the accumulator registers (conv_common.hpp: a tile's bias / residual / old values are requested together;
    // the per-element form this replaces paid one dependent global-load round trip per output element)
    if ((STS_EXP & 32) && acc[0][0][0] != 12345.f) return;
    tile_epilogue<MW, NW>(a, acc, mbase, n0 + wn * NW * 32, l31, half, n_count, out_len, out_base, phase, b);
}

template <int MW, int NW, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv_mfma_kernel(ConvArgs a, int mtiles, int nx, int ny) {
    const TileId t = map_tile(nx, ny, a.B);
    if (!t.valid) return;
    conv_mfma_body<MW, NW, WM, WN>(a, mtiles, t.bx, t.by, t.bz);
}

// Grouped launch: up to kMaxGroup independent convs of identical geometry (the nResK ResBlock chains of
// one decoder stage: same channels and length, different kernel size / dilation / weights / buffers) in
// ONE grid.  z = group * B + utterance; the members are ordered longest K loop first, so the
// short-K workgroups backfill the CUs the long ones still occupy, and one member's output drain
// overlaps another member's MFMA phase.
template <int MW, int NW, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv_mfma_group_kernel(ConvGroup G, int mtiles, int B, int nx, int ny) {
    const TileId t = map_tile(nx, ny, B * G.n);
    if (!t.valid) return;
    const int gi = t.bz / B;
    // G sits at offset 0 of the kernarg segment; indexing it through the segment pointer keeps the member
    // selection a scalar load (indexing the by-value parameter would spill the whole struct to scratch)
    const ConvArgs* ga = (const ConvArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    conv_mfma_body<MW, NW, WM, WN, STS_RA_GROUP>(ga[gi], mtiles, t.bx, t.by, t.bz - gi * B);
}

// ------------------------------------------------------------------------------------------------
// Winograd-domain conv: segmented F(2,3) on the matrix cores (fewer MFMAs for the same fp32 result)
//
// This single-conv kernel is reachable through sts_debug_conv1d(mode 12) (tests, tools/conv_bench.py); the product
// path uses the same arithmetic inside the fused layer kernel below (resblock_wino_kernel).  Measured on MI355X
// (docs/HISTORY.md 5): as a single conv it beats the direct kernel by 11-21 % on the 128/64/32-channel decoder shapes.
//
// A dilated k-tap conv computes the output pair (y[n], y[n+d]) from x[n + j d], j = 0..k.  Cutting the taps
// into 3-tap (and 2-tap) segments and applying the minimal-filtering identity F(2,3) to each segment,
//     M0 += U0 (X0 - X2)   M1 += U1 (X1 + X2)   M2 += U2 (X2 - X1)   M3 += U3 (X1 - X3)
//     y[n] = M0 + M1 + M2                      y[n+d] = M1 - M2 - M3
// with X_m = x[n + (j0 + m) d] and U = G g precomputed per segment at load time (wino_pack), needs 4 (3) matrix
// products per segment and output pair where the direct form needs 6 (4): k = 3 / 7 / 11 -> 4 / 10 / 15
// MFMA sets instead of 6 / 14 / 22.  Only the well-conditioned points {0, 1, -1, inf} are used, so the fp32
// error is ~1.2x the direct kernel's (checked against fp64 in tests); sums over segments and input channels
// happen in the Winograd domain (the output transform is linear), i.e. four accumulators per output tile.
//
// Geometry: a wave owns 32 rows x 30 output PAIRS = 60 consecutive positions (30 = lcm of the dilations 1,
// 2, 3, 5, 6, 10, 15; lanes 30/31 of the 32-wide MFMA tile idle), lane l = d q + r holds the pair
// (2 d q + r, 2 d q + r + d).  The staged input row is stored de-interleaved -- position p = 2 d q + r + e d
// lives at LDS index e H + d q + r -- so the fragment of X_j is ONE contiguous LDS read at
// (j & 1) H + 30 wave + l + d (j >> 1), conflict-free.  K loop steps = (16-channel chunk, 8-channel half,
// segment): 16 (12) MFMAs per step, A fragments through a 3-deep register ring, X values 2-deep.
// ------------------------------------------------------------------------------------------------
constexpr int WINO_NT = 240;          // output positions per workgroup (4 waves x 60)
constexpr int WINO_H = 160;           // half-row length of the de-interleaved staged row
constexpr int WINO_LD = 2 * WINO_H;

template <int DUMMY>
__device__ __forceinline__ void conv_wino_body(const ConvArgs& a, const int bx, const int by, const int b) {
    constexpr int NTHR = 256;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int in_len = seg_len(a.in_seg, b);
    const int out_len = seg_len(a.out_seg, b);
    const int n0 = bx * WINO_NT;
    if (n0 >= out_len) return;
    const int m0 = by * 32;
    const size_t in_base = (size_t)seg_start(a.in_seg, b), out_base = (size_t)seg_start(a.out_seg, b);
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int d = a.tap_step, k = a.ntap;
    const int n3 = a.wino_n3, nseg = a.wino_n3 + a.wino_n2;
    const int Wneed = WINO_NT + (k - 1) * d;          // staged positions [w0, w0 + Wneed)
    const int w0 = n0 + a.tap_off;

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;

    const int nchunk = a.Cin_pad / CK;
    const int nsteps = nchunk * 2 * nseg;

    // A fragments: U[seg][xi][cin][cout]; one fragment (4 channel pairs of one xi) per sub-step
    const rsrc_t wrs = make_rsrc(a.wu, (unsigned)((size_t)nseg * 4 * a.Cin_pad * a.Cout_pad * 4));
    const unsigned a_voff = (unsigned)(((size_t)half * a.Cout_pad + m0 + l31) * 4);
    const unsigned xi_stride = (unsigned)a.Cin_pad * (unsigned)a.Cout_pad * 4u;
    auto load_a = [&](int c, int h, int sg, int xi, float (&dst)[4]) {
        const unsigned sbase = ((unsigned)sg * 4u + (unsigned)xi) * xi_stride + (unsigned)((c * CK + h * 8) * a.Cout_pad) * 4u;
#pragma unroll
        for (int p = 0; p < 4; p++)
            dst[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                wrs, (int)a_voff, (int)(sbase + (unsigned)(2 * p) * (unsigned)a.Cout_pad * 4u), 0));
    };
    // X values of segment sg: X_m = staged row at tap j0 + m (one contiguous LDS read per (m, channel pair))
    float X[4][4];
    auto load_xv = [&](int bufi, int h, int sg) {
        const int j0 = sg < n3 ? 3 * sg : 3 * n3 + 2 * (sg - n3);
        const float* sb = smem + bufi * (CK * WINO_LD) + (h * 8 + half) * WINO_LD + 30 * wn + l31;
#pragma unroll
        for (int m = 0; m < 4; m++) {
            const int j = j0 + m;
            const float* sm = sb + (j & 1) * WINO_H + d * (j >> 1);
#pragma unroll
            for (int p = 0; p < 4; p++) X[m][p] = sm[2 * p * WINO_LD];
        }
    };

    // staging: thread t owns window positions t and t + 256; a chunk is staged in two 8-row halves so that
    // only 16 registers hold in-flight input (the accumulators already take 64)
    unsigned xoff[2]; int lidx[2]; bool act1;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int p = tid + i * NTHR;
        const int pos = w0 + p;
        const bool act = p < Wneed;
        if (i == 1) act1 = act;
        xoff[i] = (act && pos >= 0 && pos < in_len) ? (unsigned)pos * 4u : kOOB;
        const int q = p / (2 * d), rem = p - q * 2 * d, e = rem >= d ? 1 : 0;
        lidx[i] = e * WINO_H + d * q + (rem - e * d);
    }
    float xr[8][2];
    auto load_x = [&](int c, int hh) {          // rows hh*8 .. hh*8+7 of chunk c
#pragma unroll
        for (int i = 0; i < 2; i++)
            if (i == 0 || act1) {
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int ci = c * CK + hh * 8 + r;
                    xr[r][i] = buf_load(make_rsrc(a.x + (size_t)ci * a.x_ld + in_base, ci < a.Cin ? (unsigned)in_len * 4u : 0u), xoff[i]);
                }
            }
    };
    auto store_x = [&](int bufi, int hh) {
        float* sb = smem + bufi * (CK * WINO_LD) + hh * 8 * WINO_LD;
#pragma unroll
        for (int i = 0; i < 2; i++)
            if (i == 0 || act1) {
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    float v = xr[r][i];
                    if (a.in_act) v = v < 0.f ? v * a.in_slope : v;
                    sb[r * WINO_LD + lidx[i]] = v;
                }
            }
    };

    // ---- K loop: runtime loop over (chunk, half, segment) steps, four static sub-steps (xi) each.  The A ring
    // slot IS xi; a fragment is requested three sub-steps before its MFMAs.  X is single-buffered: the next
    // segment's values are read from LDS as soon as V3 has consumed the current ones, under xi = 3's MFMAs.
    float fa[4][4];
    int sc = 0, sh = 0, ss = 0;
    load_x(0, 0); store_x(0, 0);
    load_x(0, 1); store_x(0, 1);
    load_a(0, 0, 0, 0, fa[0]);
    load_a(0, 0, 0, 1, fa[1]);
    load_a(0, 0, 0, 2, fa[2]);
    __syncthreads();
    load_xv(0, 0, 0);
    if (nchunk > 1) load_x(1, 0);
    for (int s = 0; s < nsteps; s++) {
        int ns = ss + 1, nh = sh, nc = sc;
        if (ns == nseg) { ns = 0; nh = sh + 1; if (nh == 2) { nh = 0; nc = sc + 1; } }
        const bool more = s + 1 < nsteps;
        const bool three = ss < n3;
        float v[4];
        // xi = 0
        if (three) load_a(sc, sh, ss, 3, fa[3]);
#pragma unroll
        for (int p = 0; p < 4; p++) v[p] = X[0][p] - X[2][p];
#pragma unroll
        for (int p = 0; p < 4; p++) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][p], v[p], acc[0], 0, 0, 0);
        // xi = 1
        if (more) load_a(nc, nh, ns, 0, fa[0]);
#pragma unroll
        for (int p = 0; p < 4; p++) v[p] = X[1][p] + X[2][p];
#pragma unroll
        for (int p = 0; p < 4; p++) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][p], v[p], acc[1], 0, 0, 0);
        // xi = 2
        if (more) load_a(nc, nh, ns, 1, fa[1]);
#pragma unroll
        for (int p = 0; p < 4; p++) v[p] = X[2][p] - X[1][p];
#pragma unroll
        for (int p = 0; p < 4; p++) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2][p], v[p], acc[2], 0, 0, 0);
        // xi = 3
        if (more) load_a(nc, nh, ns, 2, fa[2]);
#pragma unroll
        for (int p = 0; p < 4; p++) v[p] = X[1][p] - X[3][p];
        if (more) {
            if (nc != sc) {                       // chunk boundary: publish the second half of chunk nc
                store_x(nc & 1, 1);
                __syncthreads();
                if (nc + 1 < nchunk) load_x(nc + 1, 0);
            } else if (nh != sh) {                // middle of the chunk: first half of chunk sc + 1 goes out
                if (sc + 1 < nchunk) { store_x((sc + 1) & 1, 0); load_x(sc + 1, 1); }
            }
            load_xv(nc & 1, nh, ns);
        }
        if (three) {
#pragma unroll
            for (int p = 0; p < 4; p++) acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3][p], v[p], acc[3], 0, 0, 0);
        }
        ss = ns; sh = nh; sc = nc;
    }

    // ---- output transform + epilogue
    if (l31 >= 30) return;
    const int q = l31 / d, rr = l31 - q * d;
    const int na = n0 + 60 * wn + 2 * d * q + rr, nb = na + d;
    const bool va = na < out_len, vb = nb < out_len;
    static_for<0, 16>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (row < a.Cout) {
            float ya = acc[0][r] + acc[1][r] + acc[2][r];
            float yb = acc[1][r] - acc[2][r] - acc[3][r];
            float bv = a.bias ? a.bias[row] : 0.f;
            if (a.ubias) bv += a.ubias[(size_t)row * a.ubias_ld + b];
            if (va) epi_scalar(a, row, out_base + (size_t)na, ya + bv);
            if (vb) epi_scalar(a, row, out_base + (size_t)nb, yb + bv);
        }
    });
}

__global__ __launch_bounds__(256) void conv_wino_kernel(ConvArgs a, int nx, int ny) {
    const TileId t = map_tile(nx, ny, a.B);
    if (!t.valid) return;
    conv_wino_body<0>(a, t.bx, t.by, t.bz);
}

// ------------------------------------------------------------------------------------------------
// fused ResBlock layer for the narrow decoder stages (C = 32 * MW <= 64)
//   y = x + conv2_{k2,d=1}( lrelu( conv1_{k1,d1}( lrelu(x) ) ) )        (ResBlock1.cpp:55-69, one dilation)
// Timing experiments on MI355X (tools/exp_build.sh) showed that at these widths a third of a conv's
// time is its output drain (22 MB written and flushed at kernel end for ~1-4 GFLOP of work).  Here a
// workgroup produces a (C x NT) output tile from scratch: phase 1 computes the intermediate for the NT
// columns plus conv2's halo (exactly 128 columns = 4 waves x 32) with the same staged-input K loop as
// conv_mfma_body and parks it, biased and activated, in LDS; phase 2 runs conv2 straight out of that LDS
// tile (no staging, no barrier in its K loop) and adds bias and the residual in the epilogue.  The
// intermediate tensor, one launch and one kernel-end flush per layer disappear; the accumulation order
// of both convs is the one of the unfused kernels, so results are bit-identical to them.
// ------------------------------------------------------------------------------------------------
constexpr int RL_W1 = 128;            // intermediate columns per workgroup
constexpr int RL_XW = RL_W1 + 64;     // staged input window: W1 + 2 * h1, h1 = d1 (k1 - 1) / 2 <= 32

template <int MW, int WM>
__global__ __launch_bounds__(256 * WM) __attribute__((amdgpu_waves_per_eu(WM == 2 ? 4 : 1))) void resblock_layer_kernel(ResLayerGroup G, int nx) {
    // WM row groups of 32 * MW channels x 4 column tiles of 32: 4 * WM waves
    constexpr int C = 32 * MW * WM;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const t1 = smem;   // [C][RL_W1] (+ slack for masked columns); ALIASES the input staging buffers,
                              // which are dead once phase 1 has issued its last MFMA (barrier below)
    const TileId t = map_tile(nx, 1, G.B * G.n);
    if (!t.valid) return;
    const int gi = t.bz / G.B, b = t.bz - gi * G.B;
    const ResLayerArgs& a = ((const ResLayerArgs*)__builtin_amdgcn_kernarg_segment_ptr())[gi];
    const int h1 = a.dil1 * (a.k1 - 1) / 2, h2 = (a.k2 - 1) / 2;
    const int NT = RL_W1 - 2 * h2;                    // output columns of this workgroup
    const int len = seg_len(G.seg, b);
    const int n0 = t.bx * NT;
    if (n0 >= len) return;
    const size_t base = (size_t)seg_start(G.seg, b);
    const int tid = threadIdx.x, lane = tid & 63, wn = (tid >> 6) & 3, wm = tid >> 8;
    const int mbase = wm * MW * 32;
    const int l31 = lane & 31, half = lane >> 5;

    f32x16 acc[MW];
#pragma unroll
    for (int i = 0; i < MW; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;

    constexpr int nchunk = C / CK;
    const unsigned a_voff = (unsigned)((half * C + mbase + l31) * 4);
    float fa[3][CK / 2][MW], fb[2][CK / 2][1];
    int sj, sc, aj, ac;

    // ================= phase 1: t1 = lrelu(conv1(lrelu(x)) + b1) on columns [n0 - h2, n0 - h2 + 128) ==========
    {
        const rsrc_t wrs = make_rsrc(a.w1, (unsigned)(a.k1 * C * C * 4));
        auto load_a = [&](int c, int j, float (&dst)[CK / 2][MW]) {
            const unsigned sbase = (unsigned)(((j * C + c * CK) * C) * 4);
#pragma unroll
            for (int p = 0; p < CK / 2; p++)
#pragma unroll
                for (int i = 0; i < MW; i++)
                    dst[p][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        wrs, (int)a_voff, (int)(sbase + (unsigned)(2 * p * C * 4 + i * 128)), 0));
        };
        auto load_b = [&](int bufi, int j, float (&dst)[CK / 2][1]) {
            const float* sb = smem + bufi * (CK * RL_XW) + wn * 32 + l31 + j * a.dil1;
#pragma unroll
            for (int p = 0; p < CK / 2; p++) dst[p][0] = sb[(2 * p + half) * RL_XW];
        };
        const int win0 = n0 - h2 - h1;                // position of staged column 0
        const bool stager = tid < RL_XW;              // waves 0..2 stage (192 columns)
        unsigned xoff;
        {
            const int pos = win0 + tid;
            xoff = (stager && tid < RL_W1 + 2 * h1 && pos >= 0 && pos < len) ? (unsigned)pos * 4u : kOOB;
        }
        float xr[CK];
        auto load_x = [&](int c) {
            if (stager) {
#pragma unroll
                for (int r = 0; r < CK; r++)
                    xr[r] = buf_load(make_rsrc(a.x + (size_t)(c * CK + r) * G.ld + base, (unsigned)len * 4u), xoff);
            }
        };
        auto store_tile = [&](int bufi) {
            if (stager) {
                float* sb = smem + bufi * (CK * RL_XW) + tid;
#pragma unroll
                for (int r = 0; r < CK; r++) { float v = xr[r]; sb[r * RL_XW] = v < 0.f ? v * G.slope : v; }
            }
        };
        const int nsteps = nchunk * a.k1;
        sj = 0; sc = 0; aj = 0; ac = 0;
        auto request_a = [&](float (&dst)[CK / 2][MW]) {
            load_a(ac, aj, dst);
            if (++aj == a.k1) { aj = 0; ac++; }
        };
        auto do_step = [&](float (&acur)[CK / 2][MW], float (&anew)[CK / 2][MW], float (&bcur)[CK / 2][1],
                           float (&bnxt)[CK / 2][1], int s) {
            const bool last_tap = sj + 1 == a.k1;
            int nj = sj + 1, nc = sc;
            if (last_tap) { nj = 0; nc = sc + 1; }
            if (s + 2 < nsteps) request_a(anew);
            if (s + 1 < nsteps) {
                if (last_tap) {
                    store_tile(nc & 1);
                    __syncthreads();
                    if (nc + 1 < nchunk) load_x(nc + 1);
                }
                load_b(nc & 1, nj, bnxt);
            }
#pragma unroll
            for (int p = 0; p < CK / 2; p++)
#pragma unroll
                for (int i = 0; i < MW; i++)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[p][i], bcur[p][0], acc[i], 0, 0, 0);
            sj = nj; sc = nc;
        };
        load_x(0);
        request_a(fa[0]);
        if (nsteps > 1) request_a(fa[1]);
        store_tile(0);
        __syncthreads();
        load_b(0, 0, fb[0]);
        if (nchunk > 1) load_x(1);
        for (int s = 0; s < nsteps; s += 6)
            static_for<0, 6>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if (s + u < nsteps) do_step(fa[u % 3], fa[(u + 2) % 3], fb[u % 2], fb[(u + 1) % 2], s + u);
            });
    }
    // park the intermediate: bias, conv2's input activation, conv2's zero padding outside [0, len)
    __syncthreads();          // every wave is done reading the staged input (t1 overwrites it)
    {
        const int col = wn * 32 + l31;
        const int pos = n0 - h2 + col;
        const bool inside = pos >= 0 && pos < len;
        static_for<0, MW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            float b1v[16];                               // the lane's 16 bias values, requested together
#pragma unroll
            for (int r = 0; r < 16; r++) b1v[r] = a.b1 ? a.b1[mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half] : 0.f;
            static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int row = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                float v = acc[i][r];
                if (a.b1) v += b1v[r];
                v = v < 0.f ? v * G.slope : v;
                t1[row * RL_W1 + col] = inside ? v : 0.f;
                acc[i][r] = 0.f;
            });
        });
    }
    __syncthreads();

    // ================= phase 2: y = conv2(t1) + b2 + x on columns [n0, n0 + NT) ==============================
    {
        const rsrc_t wrs = make_rsrc(a.w2, (unsigned)(a.k2 * C * C * 4));
        auto load_a = [&](int c, int j, float (&dst)[CK / 2][MW]) {
            const unsigned sbase = (unsigned)(((j * C + c * CK) * C) * 4);
#pragma unroll
            for (int p = 0; p < CK / 2; p++)
#pragma unroll
                for (int i = 0; i < MW; i++)
                    dst[p][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                        wrs, (int)a_voff, (int)(sbase + (unsigned)(2 * p * C * 4 + i * 128)), 0));
        };
        auto load_b = [&](int c, int j, float (&dst)[CK / 2][1]) {
            const float* sb = t1 + (c * CK) * RL_W1 + wn * 32 + l31 + j;
#pragma unroll
            for (int p = 0; p < CK / 2; p++) dst[p][0] = sb[(2 * p + half) * RL_W1];
        };
        const int nsteps = nchunk * a.k2;
        sj = 0; sc = 0; aj = 0; ac = 0;
        auto request_a = [&](float (&dst)[CK / 2][MW]) {
            load_a(ac, aj, dst);
            if (++aj == a.k2) { aj = 0; ac++; }
        };
        auto do_step = [&](float (&acur)[CK / 2][MW], float (&anew)[CK / 2][MW], float (&bcur)[CK / 2][1],
                           float (&bnxt)[CK / 2][1], int s) {
            int nj = sj + 1, nc = sc;
            if (nj == a.k2) { nj = 0; nc = sc + 1; }
            if (s + 2 < nsteps) request_a(anew);
            if (s + 1 < nsteps) load_b(nc, nj, bnxt);
#pragma unroll
            for (int p = 0; p < CK / 2; p++)
#pragma unroll
                for (int i = 0; i < MW; i++)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(acur[p][i], bcur[p][0], acc[i], 0, 0, 0);
            sj = nj; sc = nc;
        };
        request_a(fa[0]);
        if (nsteps > 1) request_a(fa[1]);
        load_b(0, 0, fb[0]);
        for (int s = 0; s < nsteps; s += 6)
            static_for<0, 6>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if (s + u < nsteps) do_step(fa[u % 3], fa[(u + 2) % 3], fb[u % 2], fb[(u + 1) % 2], s + u);
            });
    }
    {
        const int col = wn * 32 + l31;
        const int pos = n0 + col;
        if (col < NT && pos < len) {
            const size_t opos = base + (size_t)pos;
            static_for<0, MW>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                float b2v[16], xv[16];                   // bias and residual of the lane's 16 rows, requested together
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    b2v[r] = a.b2 ? a.b2[row] : 0.f;
                    xv[r] = a.x[(size_t)row * G.ld + opos];
                }
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int row = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    float v = acc[i][r];
                    if (a.b2) v += b2v[r];
                    a.y[(size_t)row * G.ld + opos] = v + xv[r];
                }
            });
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fused ResBlock layer, both convs in the Winograd domain (resblock_layer_kernel x conv_wino_body)
//   WM row groups of 32 channels x WN column groups of 30 output pairs (60 positions): 4 accumulators per wave.
//   Phase 1 produces P = 60 WN intermediate positions (conv2's halo included) and parks them in LDS in the
//   de-interleaved layout conv2 (dilation 1) reads: position p at (p & 1) H2 + (p >> 1); phase 2 runs conv2 out of
//   that tile.  K loops: (16-channel chunk, 8-channel half, segment) steps of four static sub-steps, as in
//   conv_wino_body.
// ------------------------------------------------------------------------------------------------
template <int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(WM * WN >= 8 ? 4 : 1)))
void resblock_wino_kernel(ResLayerGroup G, int nx) {
    constexpr int C = 32 * WM, NTHR = 64 * WM * WN, P = 60 * WN;
    constexpr int H1 = 30 * WN + 40, LD1 = 2 * H1;      // staged input row (de-interleaved by conv1's dilation)
    constexpr int H2 = 30 * WN + 16, LD2 = 2 * H2;      // parked intermediate row (de-interleaved, dilation 1)
    constexpr int RI = (P + MAX_HALO + NTHR - 1) / NTHR;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const t1 = smem;                             // aliases the staging buffers (dead after phase 1)
    const TileId t = map_tile(nx, 1, G.B * G.n);
    if (!t.valid) return;
    const int gi = t.bz / G.B, b = t.bz - gi * G.B;
    const ResLayerArgs& a = ((const ResLayerArgs*)__builtin_amdgcn_kernarg_segment_ptr())[gi];
    const int d = a.dil1;
    const int h1 = d * (a.k1 - 1) / 2, h2 = (a.k2 - 1) / 2;
    const int NT = P - 2 * h2;
    const int len = seg_len(G.seg, b);
    const int n0 = t.bx * NT;
    if (n0 >= len) return;
    const size_t base = (size_t)seg_start(G.seg, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN, wm = wave / WN;
    const int m0 = wm * 32;
    const int l31 = lane & 31, half = lane >> 5;

    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    constexpr int nchunk = C / CK;
    const unsigned a_voff = (unsigned)((half * C + m0 + l31) * 4);
    constexpr unsigned xi_stride = (unsigned)(C * C * 4);
    float X[4][4], fa[4][4];

    // ================= phase 1 =================
    {
        int n3, n2; wino_split(a.k1, &n3, &n2);
        const int nseg = n3 + n2, nsteps = nchunk * 2 * nseg;
        const rsrc_t wrs = make_rsrc(a.wu1, (unsigned)(nseg * 4 * C * C * 4));
        auto load_a = [&](int c, int h, int sg, int xi, float (&dst)[4]) {
            const unsigned sbase = ((unsigned)sg * 4u + (unsigned)xi) * xi_stride + (unsigned)((c * CK + h * 8) * C) * 4u;
#pragma unroll
            for (int p = 0; p < 4; p++)
                dst[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, (int)a_voff, (int)(sbase + (unsigned)(2 * p * C * 4)), 0));
        };
        auto load_xv = [&](int bufi, int h, int sg) {
            const int j0 = sg < n3 ? 3 * sg : 3 * n3 + 2 * (sg - n3);
            const float* sb = smem + bufi * (CK * LD1) + (h * 8 + half) * LD1 + 30 * wn + l31;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int j = j0 + m;
                const float* sm = sb + (j & 1) * H1 + d * (j >> 1);
#pragma unroll
                for (int p = 0; p < 4; p++) X[m][p] = sm[2 * p * LD1];
            }
        };
        const int Wneed = P + (a.k1 - 1) * d;
        const int w0 = n0 - h2 - h1;
        unsigned xoff[RI]; int lidx[RI]; bool act[RI];
#pragma unroll
        for (int i = 0; i < RI; i++) {
            const int p = tid + i * NTHR;
            const int pos = w0 + p;
            act[i] = p < Wneed;
            xoff[i] = (act[i] && pos >= 0 && pos < len) ? (unsigned)pos * 4u : kOOB;
            const int q = p / (2 * d), rem = p - q * 2 * d, e = rem >= d ? 1 : 0;
            lidx[i] = e * H1 + d * q + (rem - e * d);
        }
        float xr[8][RI];
        auto load_x = [&](int c, int hh) {
#pragma unroll
            for (int i = 0; i < RI; i++)
                if (act[i]) {
#pragma unroll
                    for (int r = 0; r < 8; r++)
                        xr[r][i] = buf_load(make_rsrc(a.x + (size_t)(c * CK + hh * 8 + r) * G.ld + base, (unsigned)len * 4u), xoff[i]);
                }
        };
        auto store_x = [&](int bufi, int hh) {
            float* sb = smem + bufi * (CK * LD1) + hh * 8 * LD1;
#pragma unroll
            for (int i = 0; i < RI; i++)
                if (act[i]) {
#pragma unroll
                    for (int r = 0; r < 8; r++) { const float v = xr[r][i]; sb[r * LD1 + lidx[i]] = v < 0.f ? v * G.slope : v; }
                }
        };
        int sc = 0, sh = 0, ss = 0;
        load_x(0, 0); store_x(0, 0);
        load_x(0, 1); store_x(0, 1);
        load_a(0, 0, 0, 0, fa[0]);
        load_a(0, 0, 0, 1, fa[1]);
        load_a(0, 0, 0, 2, fa[2]);
        __syncthreads();
        load_xv(0, 0, 0);
        if (nchunk > 1) load_x(1, 0);
        for (int s = 0; s < nsteps; s++) {
            int ns = ss + 1, nh = sh, nc = sc;
            if (ns == nseg) { ns = 0; nh = sh + 1; if (nh == 2) { nh = 0; nc = sc + 1; } }
            const bool more = s + 1 < nsteps;
            const bool three = ss < n3;
            float v[4];
            load_a(sc, sh, ss, 3, fa[3]);      // unconditional (a 2-tap segment's fourth slab is zeros and is not multiplied)
#pragma unroll
            for (int p = 0; p < 4; p++) v[p] = X[0][p] - X[2][p];
#pragma unroll
            for (int p = 0; p < 4; p++) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][p], v[p], acc[0], 0, 0, 0);
            load_a(nc, nh, ns, 0, fa[0]);      // past the last step these read inside the descriptor and are never used
#pragma unroll
            for (int p = 0; p < 4; p++) v[p] = X[1][p] + X[2][p];
#pragma unroll
            for (int p = 0; p < 4; p++) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][p], v[p], acc[1], 0, 0, 0);
            load_a(nc, nh, ns, 1, fa[1]);
#pragma unroll
            for (int p = 0; p < 4; p++) v[p] = X[2][p] - X[1][p];
#pragma unroll
            for (int p = 0; p < 4; p++) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2][p], v[p], acc[2], 0, 0, 0);
            load_a(nc, nh, ns, 2, fa[2]);
#pragma unroll
            for (int p = 0; p < 4; p++) v[p] = X[1][p] - X[3][p];
            if (more) {
                if (nc != sc) {
                    store_x(nc & 1, 1);
                    __syncthreads();
                    if (nc + 1 < nchunk) load_x(nc + 1, 0);
                } else if (nh != sh) {
                    if (sc + 1 < nchunk) { store_x((sc + 1) & 1, 0); load_x(sc + 1, 1); }
                }
                load_xv(nc & 1, nh, ns);
            }
            if (three) {
#pragma unroll
                for (int p = 0; p < 4; p++) acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3][p], v[p], acc[3], 0, 0, 0);
            }
            ss = ns; sh = nh; sc = nc;
        }
    }
    // park the intermediate (output transform, bias, conv2's input activation, conv2's zero padding)
    __syncthreads();
    {
        const int q = l31 / d, rr = l31 - q * d;
        const int pa = 60 * wn + 2 * d * q + rr, pb = pa + d;
        const int ga = n0 - h2 + pa, gb = n0 - h2 + pb;
        const bool ia = ga >= 0 && ga < len, ib = gb >= 0 && gb < len;
        if (l31 < 30) {
            float b1v[16];                               // the lane's 16 bias values, requested together
#pragma unroll
            for (int r = 0; r < 16; r++) b1v[r] = a.b1 ? a.b1[m0 + (r & 3) + 8 * (r >> 2) + 4 * half] : 0.f;
            static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int row = m0 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const float bv = b1v[r];
                float ya = acc[0][r] + acc[1][r] + acc[2][r] + bv;
                float yb = acc[1][r] - acc[2][r] - acc[3][r] + bv;
                ya = ya < 0.f ? ya * G.slope : ya;
                yb = yb < 0.f ? yb * G.slope : yb;
                t1[row * LD2 + (pa & 1) * H2 + (pa >> 1)] = ia ? ya : 0.f;
                t1[row * LD2 + (pb & 1) * H2 + (pb >> 1)] = ib ? yb : 0.f;
            });
        }
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][r] = 0.f;
    }
    __syncthreads();

    // ================= phase 2 =================
    {
        int n3, n2; wino_split(a.k2, &n3, &n2);
        const int nseg = n3 + n2, nsteps = nchunk * 2 * nseg;
        const rsrc_t wrs = make_rsrc(a.wu2, (unsigned)(nseg * 4 * C * C * 4));
        auto load_a = [&](int c, int h, int sg, int xi, float (&dst)[4]) {
            const unsigned sbase = ((unsigned)sg * 4u + (unsigned)xi) * xi_stride + (unsigned)((c * CK + h * 8) * C) * 4u;
#pragma unroll
            for (int p = 0; p < 4; p++)
                dst[p] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrs, (int)a_voff, (int)(sbase + (unsigned)(2 * p * C * 4)), 0));
        };
        auto load_xv = [&](int c, int h, int sg) {
            const int j0 = sg < n3 ? 3 * sg : 3 * n3 + 2 * (sg - n3);
            const float* sb = t1 + (c * CK + h * 8 + half) * LD2 + 30 * wn + l31;
#pragma unroll
            for (int m = 0; m < 4; m++) {
                const int j = j0 + m;
                const float* sm = sb + (j & 1) * H2 + (j >> 1);
#pragma unroll
                for (int p = 0; p < 4; p++) X[m][p] = sm[2 * p * LD2];
            }
        };
        int sc = 0, sh = 0, ss = 0;
        load_a(0, 0, 0, 0, fa[0]);
        load_a(0, 0, 0, 1, fa[1]);
        load_a(0, 0, 0, 2, fa[2]);
        load_xv(0, 0, 0);
        for (int s = 0; s < nsteps; s++) {
            int ns = ss + 1, nh = sh, nc = sc;
            if (ns == nseg) { ns = 0; nh = sh + 1; if (nh == 2) { nh = 0; nc = sc + 1; } }
            const bool more = s + 1 < nsteps;
            const bool three = ss < n3;
            float v[4];
            load_a(sc, sh, ss, 3, fa[3]);      // unconditional (a 2-tap segment's fourth slab is zeros and is not multiplied)
#pragma unroll
            for (int p = 0; p < 4; p++) v[p] = X[0][p] - X[2][p];
#pragma unroll
            for (int p = 0; p < 4; p++) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[0][p], v[p], acc[0], 0, 0, 0);
            load_a(nc, nh, ns, 0, fa[0]);      // past the last step these read inside the descriptor and are never used
#pragma unroll
            for (int p = 0; p < 4; p++) v[p] = X[1][p] + X[2][p];
#pragma unroll
            for (int p = 0; p < 4; p++) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[1][p], v[p], acc[1], 0, 0, 0);
            load_a(nc, nh, ns, 1, fa[1]);
#pragma unroll
            for (int p = 0; p < 4; p++) v[p] = X[2][p] - X[1][p];
#pragma unroll
            for (int p = 0; p < 4; p++) acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[2][p], v[p], acc[2], 0, 0, 0);
            load_a(nc, nh, ns, 2, fa[2]);
#pragma unroll
            for (int p = 0; p < 4; p++) v[p] = X[1][p] - X[3][p];
            load_xv(more ? nc : 0, nh, ns);
            if (three) {
#pragma unroll
                for (int p = 0; p < 4; p++) acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[3][p], v[p], acc[3], 0, 0, 0);
            }
            ss = ns; sh = nh; sc = nc;
        }
    }
    if (l31 < 30) {
        const int oa = 60 * wn + 2 * l31, ob = oa + 1;
        const bool va = oa < NT && n0 + oa < len, vb = ob < NT && n0 + ob < len;
        const size_t opos = base + (size_t)(n0 + oa);
        // bias and residual requested together, 8 rows x 2 positions at a time (the kernel sits at its 128-register budget:
        // 4 waves per SIMD -- a batch of all 16 rows would cost that)
        static_for<0, 2>([&](auto hc) {
            constexpr int hb = decltype(hc)::value * 8;
            float b2v[8], xa[8], xb[8];
#pragma unroll
            for (int r = 0; r < 8; r++) {
                const int row = m0 + ((hb + r) & 3) + 8 * ((hb + r) >> 2) + 4 * half;
                const size_t o = (size_t)row * G.ld + opos;
                b2v[r] = a.b2 ? a.b2[row] : 0.f;
                xa[r] = va ? a.x[o] : 0.f;
                xb[r] = vb ? a.x[o + 1] : 0.f;
            }
            static_for<0, 8>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                const int row = m0 + ((hb + r) & 3) + 8 * ((hb + r) >> 2) + 4 * half;
                const float bv = b2v[r];
                const float ya = acc[0][hb + r] + acc[1][hb + r] + acc[2][hb + r] + bv;
                const float yb = acc[1][hb + r] - acc[2][hb + r] - acc[3][hb + r] + bv;
                const size_t o = (size_t)row * G.ld + opos;
                if (va) a.y[o] = ya + xa[r];
                if (vb) a.y[o + 1] = yb + xb[r];
            });
        });
    }
}

// ------------------------------------------------------------------------------------------------
// split-K matrix-core kernel for LATENCY-bound shapes (batch-1 text encoder / flow / duration
// predictor: N = 128..700 positions, a few dozen output tiles, K up to 2304).  The LDS-staged kernel
// above would run such a conv on a handful of CUs as one long dependent chain of global-load round
// trips.  Here every workgroup owns ONE (32*MW) x (32*NW) output tile and its KS waves split the K
// loop (groups of 8 input channels of one tap, interleaved over the waves); both operands stream
// straight from L2 into a 3-deep register ring -- no LDS staging, no barrier inside the K loop -- and
// the KS partial accumulators are summed through LDS once, followed by the same fused epilogues.
// ------------------------------------------------------------------------------------------------
template <int MW, int NW>
__global__ __launch_bounds__(1024) void conv_mfma_splitk_kernel(ConvArgs a, int mtiles) {
    constexpr int G = 4, E = MW * NW * 16;
    constexpr int D = MW * NW == 1 ? 6 : (MW * NW == 2 ? 4 : 3);   // register ring depth (groups in flight)
    extern __shared__ __attribute__((aligned(16))) float red[];   // [KS][E][64]
    const int KS = blockDim.x >> 6;
    const int nsl = a.kslices > 1 ? a.kslices : 1;
    const int b = blockIdx.z / nsl, slice = blockIdx.z - b * nsl;
    const int orig_len = seg_len(a.in_seg, b);
    const int in_len = orig_len + (a.in_reflect ? 1 : 0);
    const int out_len = seg_len(a.out_seg, b);
    const int n_count = a.transposed ? in_len + a.n_extra : out_len;
    const int n0 = blockIdx.x * 32 * NW;
    if (n0 >= n_count) return;
    const int phase = blockIdx.y / mtiles;
    const int m0 = (blockIdx.y - phase * mtiles) * 32 * MW;
    const size_t in_base = (size_t)seg_start(a.in_seg, b), out_base = (size_t)seg_start(a.out_seg, b);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const float* w = a.w + (size_t)phase * a.ntap * a.Cin_pad * a.Cout_pad;
    bool mvalid[MW];
#pragma unroll
    for (int i = 0; i < MW; i++) mvalid[i] = (m0 + i * 32) < a.Cout_pad;

    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; i++)
#pragma unroll
        for (int q = 0; q < NW; q++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][q][r] = 0.f;

    // K is walked in groups of G = 4 channel pairs (8 input channels of one tap); wave w takes groups
    // w, w + KS, ...  The (tap, channel) position of a wave's next group is tracked incrementally in
    // SCALAR registers (no integer division, no 64-bit vector address arithmetic): both operands go
    // through raw buffer descriptors with the uniform part of the address in the scalar offset.
    const int gpt = a.Cin_pad / (2 * G);   // groups per tap
    const int gall = a.ntap * gpt;
    const int gper = (gall + nsl - 1) / nsl;
    const int gfirst = slice * gper;                                   // this workgroup's share of K: groups [gfirst, ngroups)
    const int ngroups = (STS_EXP & 64) ? gfirst : (gfirst + gper < gall ? gfirst + gper : gall);   // EXP 64: no K loop at all
    float ra[D][G][MW], rb[D][G][NW];
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const int sKS = __builtin_amdgcn_readfirstlane(KS);
    // x: one descriptor based at this utterance's first sample, rows folded into the offset
    // (the launcher only picks this kernel while Cin_pad * x_ld * 4 fits the 32-bit offset)
    const rsrc_t xrs = make_rsrc(a.x + in_base, (unsigned)(((size_t)(a.Cin - 1) * a.x_ld + orig_len) * 4));   // < kOOB
    const rsrc_t wrs = make_rsrc(w, (unsigned)((size_t)a.ntap * a.Cin_pad * a.Cout_pad * 4));
    const unsigned ld4 = (unsigned)a.x_ld * 4u;
    const unsigned x_lane = (unsigned)half * ld4;                                       // odd channel of a pair
    const unsigned a_lane = (unsigned)(((size_t)half * a.Cout_pad + m0 + l31) * 4);
    const int lanepos = n0 + l31;
    int lj = (gfirst + swave) / gpt, lc = (gfirst + swave - lj * gpt) * 2 * G;     // (tap, first channel) of the next group to LOAD
    const int jstep = sKS / gpt, cstep = (sKS - jstep * gpt) * 2 * G;

    auto load_group = [&](float (&fa)[G][MW], float (&fb)[G][NW]) {
        const int shift = lj * a.tap_step + a.tap_off;
        unsigned off[NW];
#pragma unroll
        for (int q = 0; q < NW; q++) {
            int pos = lanepos + q * 32 + shift;
            bool v = pos >= 0 && pos < in_len;
            if (a.in_reflect) { pos = pos - 1; if (pos < 0) { pos = 1; v = v && orig_len > 1; } }
            off[q] = v ? (unsigned)pos * 4u + x_lane : kOOB;
        }
        const unsigned xs0 = (unsigned)lc * ld4;                                          // scalar
        const unsigned ws0 = (unsigned)((((size_t)lj * a.Cin_pad + lc) * a.Cout_pad) * 4);   // scalar
#pragma unroll
        for (int i = 0; i < G; i++) {
            // channels lc + 2i (+ half).  The scalar offset is NOT part of the hardware range check, so padded
            // channels (>= Cin) are masked through the vector offset explicitly.
            const bool cv = lc + 2 * i + half < a.Cin;
#pragma unroll
            for (int q = 0; q < NW; q++)
                fb[i][q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, (int)(cv ? off[q] : kOOB), (int)(xs0 + (unsigned)(2 * i) * ld4), 0));
#pragma unroll
            for (int k = 0; k < MW; k++)
                fa[i][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    wrs, (int)a_lane, (int)(ws0 + (unsigned)(2 * i) * (unsigned)a.Cout_pad * 4u + (unsigned)k * 128u), 0));
        }
        lj += jstep; lc += cstep;
        if (lc >= a.Cin_pad) { lc -= a.Cin_pad; lj++; }
    };

    int gnext = gfirst + swave;
#pragma unroll
    for (int d = 0; d < D; d++) { if (gnext < ngroups) load_group(ra[d], rb[d]); gnext += sKS; }
    int gcur = gfirst + swave;
    while (gcur < ngroups) {
#pragma unroll
        for (int d = 0; d < D; d++) {
            if (gcur < ngroups) {
#pragma unroll
                for (int i = 0; i < G; i++) {
                    float bq[NW];
#pragma unroll
                    for (int q = 0; q < NW; q++) {
                        float v = rb[d][i][q];
                        if (a.in_act) v = v < 0.f ? v * a.in_slope : v;
                        bq[q] = v;
                    }
#pragma unroll
                    for (int k = 0; k < MW; k++)
#pragma unroll
                        for (int q = 0; q < NW; q++)
                            acc[k][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[d][i][k], bq[q], acc[k][q], 0, 0, 0);
                }
                if (gnext < ngroups) load_group(ra[d], rb[d]);
            }
            gnext += sKS; gcur += sKS;
        }
    }

    // ---- combine the KS partial tiles
    static_for<0, MW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<0, NW>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            static_for<0, 16>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                red[((size_t)wave * E + (i * NW + q) * 16 + r) * 64 + lane] = acc[i][q][r];
            });
        });
    });
    __syncthreads();
    const int out_off = a.out_off + phase;
    const bool gate = a.epi == EPI_GATE;
    // gate: tile rows (c, c + 16) = accumulator rows (r, r + 8) are the (tanh, sigmoid) pair of one channel
    for (int e = wave; e < E; e += KS) {
        const int i = e / (NW * 16);
        const int q = (e >> 4) % NW, r = e & 15;
        if (gate && r >= 8) continue;
        const int n = n0 + q * 32 + l31;
        const int pos = n * a.out_stride + out_off;
        if (n >= n_count || pos < 0 || pos >= out_len) continue;
        const size_t opos = out_base + (size_t)pos;
        const int rowp = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = 0.f, v2 = 0.f;
        for (int k = 0; k < KS; k++) {
            v += red[((size_t)k * E + (i * NW + q) * 16 + r) * 64 + lane];
            if (gate) v2 += red[((size_t)k * E + (i * NW + q) * 16 + r + 8) * 64 + lane];
        }
        if (gate) {
            if (!mvalid[i]) continue;
            if (a.bias) { v += a.bias[rowp]; v2 += a.bias[rowp + 16]; }
            if (a.ubias) { v += a.ubias[(size_t)rowp * a.ubias_ld + b]; v2 += a.ubias[(size_t)(rowp + 16) * a.ubias_ld + b]; }
            const int ch = (rowp >> 5) * 16 + (rowp & 15);
            if (ch < a.H) a.y[(size_t)ch * a.y_ld + opos] = tanh_ref(v) * sigmoid_ref(v2);
        } else {
            if (rowp >= a.Cout) continue;
            if (nsl > 1) {      // partial sum of one K slice (EPI_STORE only): bias on slice 0, the consumer adds the slices
                if (slice == 0) { if (a.bias) v += a.bias[rowp]; if (a.ubias) v += a.ubias[(size_t)rowp * a.ubias_ld + b]; }
                a.y[(size_t)slice * a.kslice_stride + (size_t)rowp * a.y_ld + opos] = v;
                continue;
            }
            if (a.bias) v += a.bias[rowp];
            if (a.ubias) v += a.ubias[(size_t)rowp * a.ubias_ld + b];
            epi_scalar(a, rowp, opos, v);
        }
    }
}

struct TileCfg { int MW, NW, WM, WN; };
static const TileCfg kTiles[] = {
    {2, 2, 2, 2},  // 0: 128 x 128
    {2, 2, 1, 4},  // 1:  64 x 256
    {1, 4, 1, 4},  // 2:  32 x 512
    {2, 1, 1, 4},  // 3:  64 x 128
    {1, 1, 1, 4},  // 4:  32 x 128
    {1, 2, 1, 4},  // 5:  32 x 256
};
constexpr int kNumTiles = 6;

bool conv_mfma_eligible(const ConvArgs& a) {
    if (a.depthwise) return false;
    // narrow outputs (e.g. the 29-row spline-parameter projection) still go to the matrix cores: the
    // weights are zero-padded to a 32-row tile and rows >= Cout are dropped in the epilogue
    if (a.Cin < 32) return false;
    if (a.Cout < 8 && a.max_n >= 4096) return false;   // long single-channel FIRs: conv_cout1_kernel is the better fit
    if (a.Cin_pad % CK != 0 || a.Cout_pad % 32 != 0) return false;
    int first = a.tap_off, last = a.tap_off + (a.ntap - 1) * a.tap_step;
    int halo = first < last ? last - first : first - last;
    if (halo > MAX_HALO) return false;
    if (a.epi == EPI_GATE && !a.gate_perm) return false;
    if (a.epi == EPI_TANH_PCM) return false;
    return true;
}

// Tile choice, from tools/conv_bench.py on MI355X (profiles/r01_conv_microbench.log).  All candidates are
// 128 columns wide (4 waves x 32 columns); a wave owns a 64-row strip (A fragment reused over two row
// tiles) when the output rows fill it and the launch still yields >= 2 workgroups per CU, else a 32-row
// strip; below one workgroup per CU the split-K kernel takes over (launch_splitk).  The gated WaveNet conv carries its
// (tanh, sigmoid) pairs inside every 32-row tile, so it takes part in the same choice.
static int pick_tile(const ConvArgs& a, int nphase) {
    const long nt = (a.max_n + 127) / 128;
    if (a.Cout_pad % 64 == 0 && nt * (a.Cout_pad / 64) * nphase * a.B >= 512) return 3;
    return 4;
}

template <int MW, int NW, int WM, int WN>
static void launch_mfma(const ConvArgs& a, int nphase, hipStream_t st) {
    constexpr int MT = 32 * MW * WM, NT = 32 * NW * WN;
    const int mt = (a.Cout_pad + MT - 1) / MT;
    const int nx = (a.max_n + NT - 1) / NT, ny = mt * nphase;
    size_t lds = (size_t)2 * CK * (NT + MAX_HALO) * sizeof(float);   // double-buffered tile
    hipLaunchKernelGGL((conv_mfma_kernel<MW, NW, WM, WN>), dim3(mapped_grid(nx, ny, a.B)), dim3(WM * WN * 64), lds, st, a, mt, nx, ny);
}

template <int MW, int NW>
static void launch_splitk(const ConvArgs& a, int nphase, hipStream_t st) {
    const int mt = (a.Cout_pad + 32 * MW - 1) / (32 * MW);
    const int nt = (a.max_n + 32 * NW - 1) / (32 * NW);
    const int nsl = a.kslices > 1 ? a.kslices : 1;
    const long steps = ((long)a.ntap * (a.Cin_pad / 8) + nsl - 1) / nsl;          // groups of 4 channel pairs (per K slice)
    constexpr int E = MW * NW * 16;
    // LDS for the partial tiles: <= 64 KiB normally; a grid that cannot even give every CU one workgroup may take
    // 128 KiB (16 waves on a two-tile workgroup: half the dependent L2/HBM round trips per wave)
    const bool sparse = (long)mt * nt * nphase * a.B * nsl <= 256;
    const int ks_cap = E <= 16 ? 16 : (E <= 32 ? (sparse ? 16 : 8) : 4);
    // enough waves that each one issues >= ~6 groups (24 MFMA rounds), but do not drown the chip
    int ks = 1;
    while (ks < ks_cap && steps / (ks * 2) >= 6 && (long)mt * nt * nphase * a.B * nsl * ks * 2 <= 4096) ks *= 2;
    dim3 grid(nt, mt * nphase, a.B * nsl);
    size_t lds = (size_t)ks * E * 64 * sizeof(float);
    hipLaunchKernelGGL((conv_mfma_splitk_kernel<MW, NW>), grid, dim3(ks * 64), lds, st, a, mt);
}

template <int MW, int NW, int WM, int WN>
static void launch_mfma_group(const ConvGroup& G, hipStream_t st) {
    constexpr int MT = 32 * MW * WM, NT = 32 * NW * WN;
    const ConvArgs& a = G.g[0];
    const int mt = (a.Cout_pad + MT - 1) / MT;
    const int nx = (a.max_n + NT - 1) / NT;
    size_t lds = (size_t)2 * CK * (NT + MAX_HALO) * sizeof(float);
    hipLaunchKernelGGL((conv_mfma_group_kernel<MW, NW, WM, WN>), dim3(mapped_grid(nx, mt, a.B * G.n)), dim3(WM * WN * 64), lds, st, G,
                       mt, a.B, nx, mt);
}

bool resblock_layer_eligible(const ResLayerGroup& G) {
    if (G.n < 1 || G.n > kMaxGroup || (G.C != 32 && G.C != 64 && G.C != 128) || G.max_n <= 0 || G.B <= 0) return false;
    for (int i = 0; i < G.n; i++) {
        const ResLayerArgs& a = G.g[i];
        if (!(a.k1 & 1) || !(a.k2 & 1) || a.k1 < 1 || a.k2 < 1) return false;
        if (a.dil1 * (a.k1 - 1) > RL_XW - RL_W1 || a.k2 - 1 > 32) return false;
        if (a.x == a.y) return false;
    }
    return true;
}

void resblock_layer(const ResLayerGroup& Gin, hipStream_t st) {
    ResLayerGroup G = Gin;
    for (int i = 1; i < G.n; i++)                       // longest K loops first
        for (int j = i; j > 0 && G.g[j].k1 + G.g[j].k2 > G.g[j - 1].k1 + G.g[j - 1].k2; j--) {
            ResLayerArgs t = G.g[j]; G.g[j] = G.g[j - 1]; G.g[j - 1] = t;
        }
    int nx = 0;
    for (int i = 0; i < G.n; i++) {
        const int NT = RL_W1 - (G.g[i].k2 - 1);
        const int n = (G.max_n + NT - 1) / NT;
        if (n > nx) nx = n;
    }
    const size_t stage = (size_t)2 * CK * RL_XW, park = (size_t)G.C * RL_W1 + 64;
    const size_t lds = (stage > park ? stage : park) * sizeof(float);
    const dim3 grid(mapped_grid(nx, 1, G.B * G.n));
    if (G.C == 32) hipLaunchKernelGGL((resblock_layer_kernel<1, 1>), grid, dim3(256), lds, st, G, nx);
    else if (G.C == 64) hipLaunchKernelGGL((resblock_layer_kernel<2, 1>), grid, dim3(256), lds, st, G, nx);
    else hipLaunchKernelGGL((resblock_layer_kernel<2, 2>), grid, dim3(512), lds, st, G, nx);
}

bool conv_wino_eligible(const ConvArgs& a) {
    if (!a.wu || a.transposed || a.depthwise || a.in_reflect || a.out_stride != 1 || a.out_off != 0) return false;
    const int d = a.tap_step;
    if (d <= 0 || d > 6 || 30 % d != 0 || a.ntap < 2) return false;
    if (a.Cin_pad % CK != 0 || a.Cout_pad % 32 != 0 || a.Cin < 32) return false;
    if ((a.ntap - 1) * d > MAX_HALO || 121 + d * ((a.ntap + 1) / 2) >= WINO_H) return false;
    if (a.epi == EPI_GATE || a.epi == EPI_TANH_PCM) return false;
    int n3, n2; wino_split(a.ntap, &n3, &n2);
    return n3 == a.wino_n3 && n2 == a.wino_n2 && n3 >= 0;
}
static size_t wino_lds() { return (size_t)2 * CK * WINO_LD * sizeof(float); }
void conv_wino(const ConvArgs& a, hipStream_t st) {
    if (a.max_n <= 0 || a.B <= 0) return;
    const int nx = (a.max_n + WINO_NT - 1) / WINO_NT, ny = a.Cout_pad / 32;
    hipLaunchKernelGGL(conv_wino_kernel, dim3(mapped_grid(nx, ny, a.B)), dim3(256), wino_lds(), st, a, nx, ny);
}
// U = G g per segment, in double, rounded once
void wino_pack(const float* w, long s_out, long s_tap, long s_in, int Cout, int k, int Cin, int Cin_pad, int Cout_pad, float* dst) {
    int n3, n2; wino_split(k, &n3, &n2);
    const int nseg = n3 + n2;
    const size_t slab = (size_t)Cin_pad * Cout_pad;
    for (size_t i = 0; i < (size_t)nseg * 4 * slab; i++) dst[i] = 0.f;
    for (int sg = 0; sg < nseg; sg++) {
        const int j0 = sg < n3 ? 3 * sg : 3 * n3 + 2 * (sg - n3), len = sg < n3 ? 3 : 2;
        for (int o = 0; o < Cout; o++)
            for (int ci = 0; ci < Cin; ci++) {
                double g[3] = {0, 0, 0};
                for (int t = 0; t < len; t++) g[t] = (double)w[(size_t)o * s_out + (size_t)(j0 + t) * s_tap + (size_t)ci * s_in];
                const double u[4] = {g[0], 0.5 * (g[0] + g[1] + g[2]), 0.5 * (g[0] - g[1] + g[2]), g[2]};
                for (int xi = 0; xi < 4; xi++) dst[((size_t)sg * 4 + xi) * slab + (size_t)ci * Cout_pad + o] = (float)u[xi];
            }
    }
}

bool resblock_wino_eligible(const ResLayerGroup& G) {
    if (!resblock_layer_eligible(G)) return false;
    for (int i = 0; i < G.n; i++) {
        const ResLayerArgs& a = G.g[i];
        if (!a.wu1 || !a.wu2 || a.k1 < 2 || a.k2 < 2 || a.k2 > 15) return false;
        if (a.dil1 < 1 || a.dil1 > 6 || 30 % a.dil1 != 0 || (a.k1 - 1) * a.dil1 > MAX_HALO) return false;
    }
    return true;
}

template <int WM, int WN>
static void launch_resblock_wino(const ResLayerGroup& G, hipStream_t st) {
    constexpr int P = 60 * WN, LD1 = 2 * (30 * WN + 40), LD2 = 2 * (30 * WN + 16);
    int nx = 0;
    for (int i = 0; i < G.n; i++) {
        const int NT = P - (G.g[i].k2 - 1);
        const int n = (G.max_n + NT - 1) / NT;
        if (n > nx) nx = n;
    }
    const size_t stage = (size_t)2 * CK * LD1, park = (size_t)32 * WM * LD2 + 64;
    const size_t lds = (stage > park ? stage : park) * sizeof(float);
    hipLaunchKernelGGL((resblock_wino_kernel<WM, WN>), dim3(mapped_grid(nx, 1, G.B * G.n)), dim3(64 * WM * WN), lds, st, G, nx);
}

void resblock_wino(const ResLayerGroup& Gin, hipStream_t st) {
    ResLayerGroup G = Gin;
    for (int i = 1; i < G.n; i++)                       // longest K loops first
        for (int j = i; j > 0 && G.g[j].k1 + G.g[j].k2 > G.g[j - 1].k1 + G.g[j - 1].k2; j--) {
            ResLayerArgs t = G.g[j]; G.g[j] = G.g[j - 1]; G.g[j - 1] = t;
        }
    if (G.C == 32) launch_resblock_wino<1, 4>(G, st);
    else if (G.C == 64) launch_resblock_wino<2, 2>(G, st);
    else launch_resblock_wino<4, 2>(G, st);
}

bool conv_group_eligible(const ConvGroup& G) {
    if (G.n < 2 || G.n > kMaxGroup) return false;
    const ConvArgs& r = G.g[0];
    for (int i = 0; i < G.n; i++) {
        const ConvArgs& a = G.g[i];
        if (!conv_mfma_eligible(a) || a.transposed || a.epi == EPI_GATE) return false;
        if (a.Cout_pad != r.Cout_pad || a.max_n != r.max_n || a.B != r.B) return false;
    }
    return true;
}

// tile: -1 automatic, 3 / 4 force 64x128 / 32x128
void conv_mfma_group(const ConvGroup& Gin, hipStream_t st, int tile) {
    ConvGroup G = Gin;
    if (G.g[0].max_n <= 0 || G.g[0].B <= 0) return;
    for (int i = 1; i < G.n; i++)                       // longest K loop first
        for (int j = i; j > 0 && (long)G.g[j].ntap * G.g[j].Cin_pad > (long)G.g[j - 1].ntap * G.g[j - 1].Cin_pad; j--) {
            ConvArgs t = G.g[j]; G.g[j] = G.g[j - 1]; G.g[j - 1] = t;
        }
    if (tile != 1 && tile != 3 && tile != 4 && tile != 5) {
        const ConvArgs& a = G.g[0];
        const long nt = (a.max_n + 127) / 128;
        tile = (a.Cout_pad % 64 == 0 && nt * (a.Cout_pad / 64) * a.B * G.n >= 1024) ? 3 : 4;
    }
    switch (tile) {
        case 1: launch_mfma_group<2, 2, 1, 4>(G, st); break;
        case 3: launch_mfma_group<2, 1, 1, 4>(G, st); break;
        case 5: launch_mfma_group<1, 2, 1, 4>(G, st); break;
        default: launch_mfma_group<1, 1, 1, 4>(G, st); break;
    }
}

// mode: -1 automatic; 0..5 force an LDS-staged tile; 6 / 7 force the split-K kernel (NW = 1 / 2)
void conv_mfma(const ConvArgs& a, hipStream_t st, int tile) {
    int nphase = a.transposed ? a.out_stride : 1;
    if (a.max_n <= 0 || a.B <= 0) return;
    bool splitk = tile == 6 || tile == 7 || a.kslices > 1;
    int nw = tile == 7 ? 2 : 1;
    if (a.kslices > 1 && tile != 7) tile = 6;
    if (tile < 0 || tile > 7) {
        tile = pick_tile(a, nphase);
        const TileCfg& t = kTiles[tile];
        const int MT = 32 * t.MW * t.WM, NT = 32 * t.NW * t.WN;
        const long blocks = (long)((a.max_n + NT - 1) / NT) * ((a.Cout_pad + MT - 1) / MT) * nphase * a.B;
        const bool fits32 = (double)a.Cin_pad * (double)a.x_ld * 4.0 < 2.0e9;    // split-K folds the rows into a 32-bit buffer offset
        if (blocks < 256 && fits32) { splitk = true; nw = ((a.max_n + 63) / 64) * ((a.Cout_pad + 31) / 32) * nphase * (long)a.B >= 512 ? 2 : 1; }
        // a short, K-heavy transposed conv (HiFi-GAN's first upsampler at one utterance: 669 positions, K = 2 x 512) leaves the
        // LDS-staged grid at ~3 long workgroups per CU; split over waves it runs 84 -> 51 us (profiles/r01_conv_microbench.log)
        else if (a.transposed && blocks < 1024 && (long)a.ntap * (a.Cin_pad / 8) >= 128 && fits32) { splitk = true; nw = 2; }
        // a 1x1 conv has no taps to share a staged window between: the LDS tile (one barrier per 16-channel chunk) is pure
        // overhead for K <= 256, at any grid size (flow res/skip conv at batch 8: 43 -> 38 us, MB-iSTFT batch 64 flow -6 %)
        else if (a.ntap == 1 && !a.transposed && a.Cin_pad <= 256 && fits32) { splitk = true; nw = 1; }
    }
    if (splitk) {
        if (nw == 2) launch_splitk<1, 2>(a, nphase, st); else launch_splitk<1, 1>(a, nphase, st);
        return;
    }
    switch (tile) {
        case 0: launch_mfma<2, 2, 2, 2>(a, nphase, st); break;
        case 1: launch_mfma<2, 2, 1, 4>(a, nphase, st); break;
        case 2: launch_mfma<1, 4, 1, 4>(a, nphase, st); break;
        case 3: launch_mfma<2, 1, 1, 4>(a, nphase, st); break;
        case 4: launch_mfma<1, 1, 1, 4>(a, nphase, st); break;
        default: launch_mfma<1, 2, 1, 4>(a, nphase, st); break;
    }
}

// ------------------------------------------------------------------------------------------------
// generic VALU kernel: one thread per (row, n); rows on grid.y, utterances on grid.z
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv_generic_kernel(ConvArgs a, int rows) {
    const int b = blockIdx.z;
    const int orig_len = seg_len(a.in_seg, b);
    const int in_len = orig_len + (a.in_reflect ? 1 : 0);
    const int out_len = seg_len(a.out_seg, b);
    const int n_count = a.transposed ? in_len + a.n_extra : out_len;
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= n_count) return;
    const int phase = blockIdx.y / rows;
    const int row = blockIdx.y - phase * rows;
    const size_t in_base = (size_t)seg_start(a.in_seg, b), out_base = (size_t)seg_start(a.out_seg, b);
    const int pos = n * a.out_stride + a.out_off + phase;
    if (pos < 0 || pos >= out_len) return;
    const size_t opos = out_base + (size_t)pos;
    const bool gate = a.epi == EPI_GATE;
    // gate: rows (c, c + H) unless the weights were tile-permuted for the matrix-core kernel
    int row2 = row + a.H;
    int rowp = row;
    if (gate && a.gate_perm) { rowp = (row >> 4) * 32 + (row & 15); row2 = rowp + 16; }
    float v = 0.f, v2 = 0.f;
    if (a.depthwise) {
        const float* xrow = a.x + (size_t)row * a.x_ld + in_base;
        for (int j = 0; j < a.ntap; j++)
            v += a.w[(size_t)j * a.Cout_pad + row] * load_in(a, xrow, n + j * a.tap_step + a.tap_off, in_len, orig_len);
    } else {
        const float* w = a.w + (size_t)phase * a.ntap * a.Cin_pad * a.Cout_pad;
        for (int j = 0; j < a.ntap; j++) {
            const int ip = n + j * a.tap_step + a.tap_off;
            if (ip < 0 || ip >= in_len) continue;
            const float* wj = w + (size_t)j * a.Cin_pad * a.Cout_pad;
            for (int ci = 0; ci < a.Cin; ci++) {
                const float xv = load_in(a, a.x + (size_t)ci * a.x_ld + in_base, ip, in_len, orig_len);
                v += wj[(size_t)ci * a.Cout_pad + rowp] * xv;
                if (gate) v2 += wj[(size_t)ci * a.Cout_pad + row2] * xv;
            }
        }
    }
    if (a.bias) { v += a.bias[rowp]; if (gate) v2 += a.bias[row2]; }
    if (a.ubias) { v += a.ubias[(size_t)rowp * a.ubias_ld + b]; if (gate) v2 += a.ubias[(size_t)row2 * a.ubias_ld + b]; }
    if (gate) a.y[(size_t)row * a.y_ld + opos] = tanh_ref(v) * sigmoid_ref(v2);
    else epi_scalar(a, row, opos, v);
}

// Cout == 1 convs over long sequences (HiFi-GAN conv_post: 32 channels x 7 taps -> 1 sample, fused
// leaky-relu in, tanh + int16 out).  The generic kernel re-reads every input 7x from L2; here a workgroup
// stages its [Cin][256 + halo] window in LDS once (coalesced rows) and the weights in LDS too.
__global__ __launch_bounds__(256) void conv_cout1_kernel(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int b = blockIdx.y;
    const int in_len = seg_len(a.in_seg, b), out_len = seg_len(a.out_seg, b);
    const int n0 = blockIdx.x * 256;
    if (n0 >= out_len) return;
    const size_t in_base = (size_t)seg_start(a.in_seg, b), out_base = (size_t)seg_start(a.out_seg, b);
    const int halo = (a.ntap - 1) * a.tap_step;            // tap_step > 0 here
    const int W = 256 + halo;
    float* xs = sm;                                        // [Cin][W]
    float* ws = sm + (size_t)a.Cin * W;                    // [ntap][Cin]
    for (int i = threadIdx.x; i < a.ntap * a.Cin; i += 256) {
        const int j = i / a.Cin, ci = i - j * a.Cin;
        ws[i] = a.w[((size_t)j * a.Cin_pad + ci) * a.Cout_pad];
    }
    // window staging: CB rows per round, all of a round's loads issued before any is used (the kernel is pure HBM streaming: one dependent
    // load per row made it latency-bound at ~0.8 TB/s; 8 rows per round -- 24 loads in flight with the three-tensor mean -- reached ~2 TB/s).
    // Round 5 measured 16 and 32 rows per round on one box (profiles/r05_ab_log.md): 33.2 us (8) / 36.6 (16) / 32.3 (32) -- nothing to gain
#ifndef STS_COUT1_CB
#define STS_COUT1_CB 8
#endif
    constexpr int CB = STS_COUT1_CB;
    // Round 6: a tile that lies wholly inside its utterance on 16-byte-aligned rows stages its 256 own columns through 16-byte loads -- a thread
    // owns 4 columns of every 4th channel, ALL its loads (Cin / 4 x nsum: 24 for conv_post) in flight at once -- and only the `halo` edge columns
    // through the scalar form.  Same values, same order of the mean: bit-identical to the scalar staging (profiles/r06_cout1_ab.log).
    const int lead = -a.tap_off;
    const bool vec = !(STS_EXP & 2048) && halo > 0 && lead >= 0 && lead <= halo && n0 + 256 <= in_len && a.Cin <= 32;   // (dword-aligned 16-byte loads)
    if (vec) {
        const int q4 = (threadIdx.x & 63) * 4, r = threadIdx.x >> 6;
        const size_t coff = in_base + (size_t)(n0 + q4);
        const float div = (float)a.nsum;
        f32x4u v[8], v1[8], v2[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const bool cv = 4 * u + r < a.Cin;
            const size_t o = coff + (size_t)(4 * u + r) * a.x_ld;
            const f32x4u z = {0.f, 0.f, 0.f, 0.f};
            v[u] = cv ? *reinterpret_cast<const f32x4u*>(a.x + o) : z;
            v1[u] = (cv && a.nsum >= 2) ? *reinterpret_cast<const f32x4u*>(a.xs1 + o) : z;
            v2[u] = (cv && a.nsum > 2) ? *reinterpret_cast<const f32x4u*>(a.xs2 + o) : z;
        }
        // the halo columns (left: lead, right: halo - lead), scalar, requested before the main block is consumed
        float ev[2] = {0.f, 0.f};
        const int nedge = halo * a.Cin;     // <= 2 * 256 in every model (halo 6 x 32 channels); more take the loop below
        int ecol[2], ech[2];
#pragma unroll
        for (int u = 0; u < 2; u++) {
            const int e = threadIdx.x + 256 * u;
            ech[u] = e / halo;
            const int ec = e - ech[u] * halo;
            ecol[u] = ec < lead ? ec : ec + 256;
            if (e < nedge) {
                const int pos = n0 + a.tap_off + ecol[u];
                if (pos >= 0 && pos < in_len) {
                    const size_t o = in_base + (size_t)pos + (size_t)ech[u] * a.x_ld;
                    float t = a.x[o];
                    if (a.nsum >= 2) { t += a.xs1[o]; if (a.nsum > 2) t += a.xs2[o]; t = t / div; }
                    ev[u] = t;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; u++)
            if (4 * u + r < a.Cin) {
                f32x4u t = v[u];
                if (a.nsum >= 2) { t = t + v1[u]; if (a.nsum > 2) t = t + v2[u]; t = t / div; }
                float* dst = xs + (size_t)(4 * u + r) * W + lead + q4;
#pragma unroll
                for (int e = 0; e < 4; e++) { float s = t[e]; if (a.in_act) s = s < 0.f ? s * a.in_slope : s; dst[e] = s; }
            }
#pragma unroll
        for (int u = 0; u < 2; u++)
            if (threadIdx.x + 256 * u < nedge) { float s = ev[u]; if (a.in_act) s = s < 0.f ? s * a.in_slope : s; xs[(size_t)ech[u] * W + ecol[u]] = s; }
        for (int e = threadIdx.x + 512; e < nedge; e += 256) {
            const int ch = e / halo, ec = e - ch * halo, col = ec < lead ? ec : ec + 256;
            const int pos = n0 + a.tap_off + col;
            float t = 0.f;
            if (pos >= 0 && pos < in_len) {
                const size_t o = in_base + (size_t)pos + (size_t)ch * a.x_ld;
                t = a.x[o];
                if (a.nsum >= 2) { t += a.xs1[o]; if (a.nsum > 2) t += a.xs2[o]; t = t / div; }
            }
            if (a.in_act) t = t < 0.f ? t * a.in_slope : t;
            xs[(size_t)ch * W + col] = t;
        }
    } else
    for (int col = threadIdx.x; col < W; col += 256) {
        const int pos = n0 + a.tap_off + col;
        const bool ok = pos >= 0 && pos < in_len;
        const size_t coff = in_base + (ok ? pos : 0);
        const float* xcol = a.x + coff;
        for (int c0 = 0; c0 < a.Cin; c0 += CB) {
            float v[CB];
#pragma unroll
            for (int u = 0; u < CB; u++) v[u] = (ok && c0 + u < a.Cin) ? xcol[(size_t)(c0 + u) * a.x_ld] : 0.f;
            if (a.nsum >= 2) {      // the input is the mean of 2 / 3 tensors (ConvArgs::nsum), formed in sum_scale's order
                float v1[CB], v2[CB];
#pragma unroll
                for (int u = 0; u < CB; u++) v1[u] = (ok && c0 + u < a.Cin) ? a.xs1[coff + (size_t)(c0 + u) * a.x_ld] : 0.f;
#pragma unroll
                for (int u = 0; u < CB; u++) v2[u] = (a.nsum > 2 && ok && c0 + u < a.Cin) ? a.xs2[coff + (size_t)(c0 + u) * a.x_ld] : 0.f;
                const float div = (float)a.nsum;
#pragma unroll
                for (int u = 0; u < CB; u++) { float t = v[u] + v1[u]; if (a.nsum > 2) t += v2[u]; v[u] = t / div; }
            }
#pragma unroll
            for (int u = 0; u < CB; u++)
                if (c0 + u < a.Cin) {
                    float t = v[u];
                    if (a.in_act) t = t < 0.f ? t * a.in_slope : t;
                    xs[(size_t)(c0 + u) * W + col] = t;
                }
        }
    }
    __syncthreads();
    const int n = n0 + threadIdx.x;
    if (n >= out_len) return;
    float v = 0.f;
    for (int j = 0; j < a.ntap; j++) {                     // same (tap, channel) order as conv_generic
        const float* xp = xs + threadIdx.x + j * a.tap_step;
        const float* wp = ws + j * a.Cin;
        for (int ci = 0; ci < a.Cin; ci++) v += wp[ci] * xp[(size_t)ci * W];
    }
    if (a.bias) v += a.bias[0];
    if (a.ubias) v += a.ubias[b];
    epi_scalar(a, 0, out_base + (size_t)n, v);
}

bool conv_cout1_takes(const ConvArgs& a) {
    if (!(a.Cout == 1 && !a.depthwise && !a.transposed && !a.in_reflect && a.tap_step > 0 && a.max_n >= 4096 && a.epi != EPI_GATE)) return false;
    return ((size_t)a.Cin * (256 + (a.ntap - 1) * a.tap_step) + (size_t)a.ntap * a.Cin) * sizeof(float) <= 64 * 1024;
}

void conv_generic(const ConvArgs& a, hipStream_t st) {
    if (a.max_n <= 0 || a.B <= 0) return;
    if (a.Cout == 1 && !a.depthwise && !a.transposed && !a.in_reflect && a.tap_step > 0 && a.max_n >= 4096 && a.epi != EPI_GATE) {
        const size_t lds = ((size_t)a.Cin * (256 + (a.ntap - 1) * a.tap_step) + (size_t)a.ntap * a.Cin) * sizeof(float);
        if (lds <= 64 * 1024) {
            hipLaunchKernelGGL(conv_cout1_kernel, dim3((a.max_n + 255) / 256, a.B), dim3(256), lds, st, a);
            return;
        }
    }
    int rows = a.epi == EPI_GATE ? a.H : a.Cout;
    int nphase = a.transposed ? a.out_stride : 1;
    dim3 grid((a.max_n + 255) / 256, rows * nphase, a.B);
    hipLaunchKernelGGL(conv_generic_kernel, grid, dim3(256), 0, st, a, rows);
}

}  // namespace sts
