// resblock_bf3.hip -- one ResBlock1 layer, x + conv2(lrelu(conv1_dilated(lrelu(x)))), of all chains of a decoder stage as ONE launch on
// the 16-bit matrix cores (split operands, conv_bf3_dev.hpp).  /root/reference/src/modules/ResBlock1.cpp:55-69.
#include "conv_bf3_dev.hpp"

namespace sts {

#ifdef STS_TILE_TRACE
int tile_trace_bind_resblock(long long* buf, unsigned capacity_records) { return tile_trace_bind(buf, capacity_records); }
#endif

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
// Waves per SIMD the register budget of the one-row-tile-per-wave forms (MW == 1: the 64- and 32-channel stages) is sized for.  Round 4:
// these forms needed 170 registers under a budget of 256 -- two more than the 168 that allow THREE waves per SIMD.  Capped at 168 (no
// spill) three workgroups share a CU instead of two and hide each other's staging / park / epilogue phases: -62 us per one-utterance step,
// -2.2 % at batch 32 (profiles/r04_ab_log.md).  Four (128 registers) spills 170-180 bytes per lane.
// Also measured in round 4 and NOT kept: a resident set of workgroups that claims tiles through per-XCD L2 counters instead of one
// workgroup per tile (the launch runs as 4-5 rounds whose boundaries idle half the chip, slot fill 0.75-0.82 in the tile trace) --
// +6 % SLOWER at one utterance, +10 % at batch 32: the tile body inlined into a loop spills 150 bytes per lane at 168 registers and
// the claim adds a barrier pair per tile, which costs more than the round boundaries it removes.
// Round 6 (profiles/r06_narrow_stage_ab.log), at 32 utterances where these two stages are 35 % of the step at 0.24-0.30 of the matrix peak:
//   * pieces compiled out (STS_RB_EXP): without its MFMAs the 64-channel launch takes 1.87 ms of 4.42, without operand loads in the K loops 3.72,
//     i.e. memory skeleton + matrix time with NO overlap although three workgroups share a CU -- the signature of the power cap:
//     overlap does not remove joules;
//   * weight fragments 3 and 4 steps ahead instead of 1 (STS_RB_ADEPTH): no difference -- the K loop is not waiting for L2;
//   * first-round workgroups of a SIMD started 4-64 us apart (lockstep hypothesis): no difference at batch, slower at one utterance;
//   * the residual requested right behind the window's loads instead of before conv2 (kept): -0.8 % at 32 utterances, 28 registers fewer.
#ifndef STS_RB_WAVES
#define STS_RB_WAVES 3
#endif
#ifndef STS_RB_ADEPTH
#define STS_RB_ADEPTH 2      // weight-fragment ring: steps in flight (round 6: 3 and 4 measured, no difference -- profiles/r06_narrow_stage_ab.log)
#endif
#ifndef STS_RB_EXP
#define STS_RB_EXP 0         // lab, TIMING only (wrong results): 1 = no B-fragment reads inside the K loops, 2 = no A-fragment loads, 4 = no MFMAs
#endif
#ifndef STS_RB_RES_LATE
#define STS_RB_RES_LATE 0     // lab: 1 = the residual requested before conv2's K loop (rounds 3-5)
#endif
template <int MW, int WM, int NW, int WN, int MATH = 0>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(MW == 1 ? STS_RB_WAVES : 2, MW == 1 ? STS_RB_WAVES : 2))) void resblock_bf3_kernel(ResLayerGroup G, int nx, int wst, int interleave) {
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
    int gi, tbx, b;
    if (interleave) { const int unit = t.bz * nx + t.bx; gi = unit % G.n; const int rest = unit / G.n; tbx = rest % nx; b = rest / nx; }
    else { gi = t.bz / G.B; tbx = t.bx; b = t.bz - gi * G.B; }
    const ResLayerArgs& a = ((const ResLayerArgs*)__builtin_amdgcn_kernarg_segment_ptr())[gi];
#ifdef STS_TILE_TRACE
    long long* tt_rec = tt_open(1, gi);       // kind 1 = fused layer (stamps: start, window staged, conv1 done, intermediate parked, conv2 done, epilogue done)
    TT_STAMP(0);
#endif
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

    // ---- weight fragments: a ring of AD steps.  Round 6: the K loops were A-LATENCY-bound -- a step of one wave is 6 MFMAs (0.09 us) and three
    // waves share a SIMD, but the tile trace showed 0.39-0.45 us per step = one L2 round trip: with the next step's fragments requested only
    // one step ahead every wave waited for them every step.  Now AD - 1 steps ahead, conv1's first fragments requested before the window is
    // staged and conv2's before the intermediate is parked.
    constexpr int AD = (MATH == 1 && MW == 1) ? STS_RB_ADEPTH : 2;      // (three-plane weights / two row tiles per wave: no registers for a deeper ring)
    const unsigned a_voff = (unsigned)lane * 16u + (unsigned)(mbase >> 5) * ABLK;
    u32x4 fa[AD][MW][NPA], fb[2][NW][NPB];
    const int nsteps1 = NCH * a.k1, nsteps2 = NCH * a.k2;
    const rsrc_t wrs1 = make_rsrc(a.wb1, (unsigned)(nsteps1 * NRT) * ABLK), wrs2 = make_rsrc(a.wb2, (unsigned)(nsteps2 * NRT) * ABLK);
    auto load_a = [&](const rsrc_t& wrs, int s, u32x4 (&dst)[MW][NPA]) {          // past the end: zeros beyond the descriptor
        const unsigned sb = (unsigned)s * ((unsigned)NRT * ABLK);
#pragma unroll
        for (int i = 0; i < MW; i++)
#pragma unroll
            for (int pl = 0; pl < NPA; pl++)
                dst[i][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)a_voff, (int)(sb + (unsigned)i * ABLK + (unsigned)(pl * 1024)), 0));
    };
#pragma unroll
    for (int r = 0; r < AD - 1; r++) load_a(wrs1, r, fa[r]);

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

    // The residual (raw x of the output columns; the staged copy is activated and split) and conv2's bias, in the transposed-quad layout of the
    // epilogue.  Round 3 requested them before conv2's K loop (they arrive under its MFMAs); round 6 requests them HERE, right behind the window's
    // own loads: by conv2 the XCD's L2 (4 MB under 32 CUs x 3 resident windows + their output tiles) has dropped the window, and the re-read went
    // to the fabric -- a third of the launch's fetch bytes (rocprofv3 FETCH_SIZE: 146-156 MB per 64-channel launch at one utterance = 1.25 x the
    // input for the windows + 1 x for the residual).  Issued now, the lines are still in flight / just filled: one fabric read serves both.
    const int lane4 = l31 & 3, m4 = (l31 >> 2) * 4;
    f32x4u xres[MW][NW][4];
    float b2v[MW][4];
#if STS_RB_RES_LATE
    auto request_residual = [&]() {
#else
    {
#endif
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
#if STS_RB_RES_LATE
    };
#else
    }
#endif

    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; i++)
#pragma unroll
        for (int q = 0; q < NW; q++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][q][r] = 0.f;
    auto mfmas = [&](u32x4 (&ac)[MW][NPA], u32x4 (&bc)[NW][NPB]) { step_mfmas<MATH, MW, NW, NPA, NPB>(acc, ac, bc); };
    __syncthreads();
    TT_STAMP(1);

    // ================= phase 1: conv1 on the P1 columns [n0 - h2, n0 - h2 + P1) =================
    {
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
        load_b(0, 0, fb[0]);
        for (int s = 0; s < nsteps1; s += 2 * AD)
            static_for<0, 2 * AD>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if (s + u < nsteps1) {
                    int nj = sj + 1, nc = sc;
                    if (nj == a.k1) { nj = 0; nc = sc + 1; }
                    if (!(STS_RB_EXP & 2)) load_a(wrs1, s + u + AD - 1, fa[(u + AD - 1) % AD]);
                    if (!(STS_RB_EXP & 1)) load_b(nc < NCH ? nc : 0, nj, fb[(u + 1) % 2]);
                    __builtin_amdgcn_sched_barrier(0);      // keep the prefetches ahead of this step's MFMAs (the scheduler would sink them)
                    if (!(STS_RB_EXP & 4)) mfmas(fa[(STS_RB_EXP & 2) ? 0 : u % AD], fb[(STS_RB_EXP & 1) ? 0 : u % 2]);
                    sj = nj; sc = nc;
                }
            });
    }
#pragma unroll
    for (int r = 0; r < AD - 1; r++) load_a(wrs2, r, fa[r]);          // conv2's first fragments travel under the park phase
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

#if STS_RB_RES_LATE
    request_residual();
#endif
    // ================= phase 2: conv2 (dilation 1) out of the parked intermediate =================
    {
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
        load_b(0, 0, fb[0]);
        for (int s = 0; s < nsteps2; s += 2 * AD)
            static_for<0, 2 * AD>([&](auto uc) {
                constexpr int u = decltype(uc)::value;
                if (s + u < nsteps2) {
                    int nj = sj + 1, nc = sc;
                    if (nj == a.k2) { nj = 0; nc = sc + 1; }
                    if (!(STS_RB_EXP & 2)) load_a(wrs2, s + u + AD - 1, fa[(u + AD - 1) % AD]);
                    if (!(STS_RB_EXP & 1)) load_b(nc < NCH ? nc : 0, nj, fb[(u + 1) % 2]);
                    __builtin_amdgcn_sched_barrier(0);      // keep the prefetches ahead of this step's MFMAs (the scheduler would sink them)
                    if (!(STS_RB_EXP & 4)) mfmas(fa[(STS_RB_EXP & 2) ? 0 : u % AD], fb[(STS_RB_EXP & 1) ? 0 : u % 2]);
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
    TT_CLOSE();
#endif
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
