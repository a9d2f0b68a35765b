// conv_h2p.hip -- two-term fp16 convs on PRE-SPLIT, channel-minor activations (round 6): the wide ResBlock stages of the decoder trunk
// (/root/reference/src/modules/ResBlock1.cpp:55-69, Generator_hifigan.cpp:151-175, nn_conv1d.cpp:118-199 are what they replace).
//
// Same arithmetic as conv_bf3_dev.hpp MATH 1 (x = hi + lo' 2^-11 as two fp16 terms, w 2^s = P0 + P1, three v_mfma_f32_32x32x16_f16 products
// per fp32 product, fp32 accumulation in the same order) -- bit-identical results -- with the split moved from the consumer's staging code to
// the producer's epilogue.  Why: the staged kernels' K loop was operand-bound (profiles/r03_f16x2_kloop_decomposition.log: 12 MFMAs = 384
// pipe cycles in a 1 070-cycle step at 11 taps, 1 550 at 3 taps): every 32 positions x 16 channels a wave staged cost 8 dword loads, ~60 VALU
// instructions (activation, two conversions, residual, scale, range check per value pair) and 2 LDS stores, per workgroup that needed the
// window, next to the weight fragments on the same vector-memory pipe.  Here
//   * activations travel in the matrix core's own operand order (kernels.hpp H2PArgs: planes / x16), written ONCE by whoever produced them:
//     a lane of a 32 x 32 accumulator tile holds 8 values of one position that are exactly one 16-byte unit of that layout (the trick of
//     resblock_bf3_kernel's parked intermediate, applied to global memory; the consumer's weights are packed in the matching k order);
//   * a consumer stages a 16-channel chunk of its window with `buffer_load_dwordx4 ... lds`: 1 KB per instruction, contiguous in memory,
//     straight into LDS, no registers, no VALU; zero padding outside the utterance is the descriptor's range check;
//   * the epilogue needs no transposes for the layouts it writes (a lane's 8 values are contiguous): residual in, x16 + planes out.
#include "conv_bf3_dev.hpp"

namespace sts {

#ifdef STS_TILE_TRACE
int tile_trace_bind_h2p(long long* buf, unsigned capacity_records) { return tile_trace_bind(buf, capacity_records); }
#endif

#define STS_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int MW, int NW, int WM, int WN>
__device__ __forceinline__ void conv_h2p_body(const H2PArgs& a, const SegView& seg, unsigned* const ovf, const int bx, const int by, const int b, [[maybe_unused]] const int tt_member) {
    constexpr int MT = 32 * MW * WM, NT = 32 * NW * WN, NWAVE = WM * WN;
    constexpr int WIN = NT + MAX_HALO;                 // staged positions per chunk
    constexpr int NSLOT = WIN / 32;                    // slots of 32 positions x 16 channels = 1 KB per plane = one LDS-DMA instruction
    constexpr int SPW = (NSLOT + NWAVE - 1) / NWAVE;   // slots per wave
    constexpr int PLANE = WIN * 32, BUF = 2 * PLANE;   // bytes
    constexpr unsigned ABLK = 2048u;                   // one (step, 32-row tile) block of the packed weights: 2 planes x 1 KB
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
#ifdef STS_TILE_TRACE
    long long* tt_rec = tt_open(2, tt_member);     // kind 2 = pre-split conv (stamps: start, first barrier, K loop done, epilogue done)
    TT_STAMP(0);
#endif
    const int len = uni(seg_len(seg, b));
    const int n0 = bx * NT;
    if (n0 >= len) return;
    const size_t base = (size_t)uni(seg_start(seg, b));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;
    const int ntap = uni(a.ntap), tap_step = uni(a.tap_step), tap_off = uni(a.tap_off);
    const int first = tap_off, last = tap_off + (ntap - 1) * tap_step;
    const int lo = first < last ? first : last, hi = first < last ? last : first;
    const int W = NT + (hi - lo);
    const int win0 = n0 + lo;
    const int mbase = by * MT + wm * MW * 32;
    const int Cin = uni(a.Cin), Cout = uni(a.Cout);
    const bool wvalid = mbase < Cout;                  // (eligibility: 32 MW divides Cout)

    f32x16 acc[MW][NW];
#pragma unroll
    for (int i = 0; i < MW; i++)
#pragma unroll
        for (int j = 0; j < NW; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    const int nchunk = Cin / CK;
    const int nsteps = nchunk * ntap;
    const int nrt = Cout / 32;
    const int nslot = (W + 31) >> 5;

    // ---- A fragments: step s = chunk * ntap + tap is one contiguous block of nrt * 2 KB (bf3_pack math 1, perm_k)
    const rsrc_t wrs = make_rsrc(a.wb, (unsigned)((size_t)nsteps * nrt * ABLK));
    const unsigned a_voff = wvalid ? (unsigned)lane * 16u + (unsigned)(mbase >> 5) * ABLK : kOOB;
    const unsigned a_step = (unsigned)nrt * ABLK;
    auto load_a = [&](int s, u32x4 (&dst)[MW][2]) {
        const unsigned sb = (unsigned)s * a_step;
#pragma unroll
        for (int i = 0; i < MW; i++)
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
                dst[i][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)a_voff, (int)(sb + (unsigned)i * ABLK + (unsigned)(pl * 1024)), 0));
    };
    // ---- B fragments of tap j out of the staged window
    const int b_t0 = wn * NW * 32 + l31 + tap_off - lo;
    auto load_b = [&](int bufi, int j, u32x4 (&dst)[NW][2]) {
        const int t = b_t0 + j * tap_step;
        const unsigned char* sb = smem3 + bufi * BUF + t * 32 + ((half ^ ((t >> 3) & 1)) << 4);
#pragma unroll
        for (int q = 0; q < NW; q++)
#pragma unroll
            for (int pl = 0; pl < 2; pl++) dst[q][pl] = *(const u32x4*)(sb + pl * PLANE + q * 1024);
    };
    // ---- window staging: slot sl of chunk c = positions [win0 + 32 sl, + 32) x 16 channels of one plane = 1 KB contiguous in memory.
    // Lane l lands at LDS byte 16 l of the slot = (column 32 sl + l / 2, unit l & 1); the unit a reader finds there must be the logical half
    // (l & 1) ^ bit 3 of the column (the XOR swizzle that keeps ds_read_b128 conflict-free for every tap shift), so the lane FETCHES that half.
    // Positions outside [0, len) lie outside the descriptor (negative offsets wrap to > 2^31): the hardware writes zeros = the conv's padding.
    unsigned dvoff[SPW];
#pragma unroll
    for (int i = 0; i < SPW; i++) {
        const int sl = swave + i * NWAVE;
        const int col = sl * 32 + (lane >> 1);
        const int pos = win0 + col;
        const bool v = pos >= 0 && pos < len;
        dvoff[i] = v ? (unsigned)pos * 32u + (unsigned)(((lane & 1) ^ ((col >> 3) & 1)) << 4) : kOOB;
    }
    const size_t in_ps = (size_t)Cin * (size_t)a.xp_ld * 2u;     // bytes of one input plane
    const unsigned char* const xp0 = (const unsigned char*)uni((const void*)a.xp) + base * 32u;
    const size_t xrow = (size_t)uni(a.xp_ld) * 32u;
    const unsigned span = (unsigned)len * 32u;
    auto dma = [&](int c, int bufi) {
#pragma unroll
        for (int pl = 0; pl < 2; pl++) {
            const rsrc_t rs = make_rsrc(xp0 + (size_t)pl * in_ps + (size_t)c * xrow, span);
#pragma unroll
            for (int i = 0; i < SPW; i++) {
                const int sl = swave + i * NWAVE;                 // wave-uniform
                if (sl < nslot)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, STS_LDS_PTR(smem3 + bufi * BUF + pl * PLANE + sl * 1024), 16, (int)dvoff[i], 0, 0, 0);
            }
        }
    };

    // ---- main loop over the steps (chunk, tap): A (L2) and B (LDS) fragments one step ahead, the window of chunk c + 1 lands while chunk c
    // is multiplied.  The compiler does not order LDS reads behind an LDS-DMA: every hand-over is an explicit s_waitcnt + barrier.
    u32x4 fa[2][MW][2], fb[2][NW][2];
    int sj = 0, sc = 0;
    auto do_step = [&](u32x4 (&acur)[MW][2], u32x4 (&anew)[MW][2], u32x4 (&bcur)[NW][2], u32x4 (&bnxt)[NW][2], int s) {
        int nj = sj + 1, nc = sc;
        if (nj == ntap) { nj = 0; nc = sc + 1; }
        const bool boundary = nc != sc && s + 1 < nsteps;
        if (boundary) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my slots of chunk nc have landed (requested a chunk ago)
            __syncthreads();                                       // everyone's have; everyone is done reading the buffer chunk nc + 1 replaces
        }
        load_a(s + 1, anew);                  // unconditional: past the last step it reads 0 beyond the descriptor, never used
        if (boundary && nc + 1 < nchunk) dma(nc + 1, (nc + 1) & 1);       // behind the weight fragments: loads return in order
        load_b(nc & 1, nj, bnxt);             // past the last step: stale LDS inside the tile, never used
        __builtin_amdgcn_sched_barrier(0);
        step_mfmas<1, MW, NW, 2, 2>(acc, acur, bcur);
        sj = nj; sc = nc;
    };
    dma(0, 0);
    if (nchunk > 1) dma(1, 1);
    load_a(0, fa[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    TT_STAMP(1);
    load_b(0, 0, fb[0]);
    // (four steps per trip through static_for: with a plain two-call body the register allocator spills the fragments of the 128 x 128 tile)
    for (int s = 0; s < nsteps; s += 4)
        static_for<0, 4>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            if (u < 2 || s + u < nsteps) { if (s + u < nsteps) do_step(fa[u % 2], fa[(u + 1) % 2], fb[u % 2], fb[(u + 1) % 2], s + u); }
        });
    TT_STAMP(2);
    if (!wvalid) return;

    // ---- epilogue.  Register hh * 8 + e of acc[i][q] = channel 16 (mbase / 16 + 2 i + hh) + 8 (e >> 2) + 4 half + (e & 3) at position
    // n0 + 32 (wn NW + q) + l31: one 16-byte unit of a plane / one 32-byte unit of an x16 tensor.
    const float ws = a.wscale;
    const float* const bias = uni(a.bias);
    const float* const res16 = uni(a.res16);
    float* const y = uni(a.y);
    float* const y16 = uni(a.y16);
    unsigned char* const yp = (unsigned char*)uni((void*)a.yp);
    const size_t out_ps = (size_t)Cout * (size_t)a.yp_ld * 2u;
    const float yslope = a.yp_slope;
    float amax = 0.f;
    const int lane4 = l31 & 3, m4 = (l31 >> 2) * 4;
    static_for<0, MW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int cc0 = (mbase >> 4) + 2 * i;
        f32x4u bv[2][2];
#pragma unroll
        for (int hh = 0; hh < 2; hh++)
#pragma unroll
            for (int u = 0; u < 2; u++)
                bv[hh][u] = bias ? *(const f32x4u*)(bias + 16 * (cc0 + hh) + 8 * u + 4 * half) : f32x4u{0.f, 0.f, 0.f, 0.f};
        static_for<0, NW>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const int ncol = n0 + (wn * NW + q) * 32;
            const int pos = ncol + l31;
            const bool ok = pos < len;
            f32x4u rv[2][2];
#pragma unroll
            for (int hh = 0; hh < 2; hh++)
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    rv[hh][u] = f32x4u{0.f, 0.f, 0.f, 0.f};
                    if (res16 && ok) rv[hh][u] = *(const f32x4u*)(res16 + (((size_t)(cc0 + hh) * a.res_ld + base + pos) * 16 + half * 8 + u * 4));
                }
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; r++) v[r] = (acc[i][q][r] * ws + bv[r >> 3][(r >> 2) & 1][r & 3]) + rv[r >> 3][(r >> 2) & 1][r & 3];
            if (y16 && ok) {
#pragma unroll
                for (int hh = 0; hh < 2; hh++)
#pragma unroll
                    for (int u = 0; u < 2; u++)
                        *(f32x4u*)(y16 + (((size_t)(cc0 + hh) * a.y16_ld + base + pos) * 16 + half * 8 + u * 4)) = f32x4u{v[hh * 8 + u * 4], v[hh * 8 + u * 4 + 1], v[hh * 8 + u * 4 + 2], v[hh * 8 + u * 4 + 3]};
            }
            if (yp && ok) {
#pragma unroll
                for (int hh = 0; hh < 2; hh++) {
                    float t[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) t[e] = __builtin_fmaxf(v[hh * 8 + e], v[hh * 8 + e] * yslope);
                    u32x4 ph, pl;
                    split8h(t, ph, pl, amax);
                    unsigned char* d = yp + (((size_t)(cc0 + hh) * a.yp_ld + base + pos) * 32 + half * 16);
                    *(u32x4*)d = ph;
                    *(u32x4*)(d + out_ps) = pl;
                }
            }
            if (y) {
                // fp32 [Cout][y_ld]: a 4 x 4 transpose inside the lane quads turns a lane's 4 consecutive rows of one column into 4
                // consecutive columns of one row -> 16-byte stores (conv_common.hpp tile_epilogue)
                const int n = ncol + m4;
                const bool full = n + 3 < len, any = n < len;
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    float w[4] = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
                    quad_transpose(w, l31);
                    if (any) {
                        float* p = y + (size_t)(mbase + i * 32 + 8 * g + 4 * half + lane4) * a.y_ld + base + n;
                        if (full) *(f32x4u*)p = f32x4u{w[0], w[1], w[2], w[3]};
                        else { for (int e = 0; e < 4; e++) if (n + e < len) p[e] = w[e]; }
                    }
                }
            }
            // one accumulator tile at a time: without the fence the scheduler hoists every tile's register reads and residual loads to
            // the front, 256+ values are live at once and the register allocator answers by spilling the K loop's operand fragments
            __builtin_amdgcn_sched_barrier(0);
        });
    });
    if (amax > kH2Limit && ovf) *ovf = 1u;
#ifdef STS_TILE_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TT_STAMP(3);
    TT_CLOSE();
#endif
}

template <int MW, int NW, int WM, int WN, int WPE>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void conv_h2p_group_kernel(H2PGroup G, int mtiles, int nx) {
    const TileId t = map_tile(nx, mtiles, G.B * G.n);
    if (!t.valid) return;
    const int gi = t.bz / G.B, b = t.bz - gi * G.B;
    const H2PArgs* ga = (const H2PArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    conv_h2p_body<MW, NW, WM, WN>(ga[gi], G.seg, G.ovf, t.bx, t.by, b, gi);
}

template <int MW, int NW, int WM, int WN, int WPE>
static void launch_h2p(const H2PGroup& G, hipStream_t st) {
    constexpr int MT = 32 * MW * WM, NT = 32 * NW * WN;
    const int mt = (G.g[0].Cout + MT - 1) / MT;
    const int nx = (G.max_n + NT - 1) / NT;
    constexpr size_t lds = (size_t)4 * (NT + MAX_HALO) * 32;
    hipLaunchKernelGGL((conv_h2p_group_kernel<MW, NW, WM, WN, WPE>), dim3(mapped_grid(nx, mt, G.B * G.n)), dim3(WM * WN * 64), lds, st, G, mt, nx);
}

bool conv_h2p_group_eligible(const H2PGroup& G) {
    if (G.n < 1 || G.n > kMaxGroup || G.max_n <= 0 || G.B <= 0) return false;
    for (int i = 0; i < G.n; i++) {
        const H2PArgs& a = G.g[i];
        if (!a.xp || !a.wb || (!a.y && !a.y16 && !a.yp)) return false;
        if (a.Cin % 16 != 0 || a.Cin < 32 || a.Cout % 128 != 0 || a.Cout != G.g[0].Cout || a.ntap < 1) return false;
        const int span = (a.ntap - 1) * a.tap_step;
        if (a.tap_step < 1 || span > MAX_HALO) return false;
        if ((double)a.xp_ld * 32.0 >= 2.0e9) return false;       // a chunk's row of positions behind one 32-bit buffer descriptor
    }
    return true;
}

// tile codes (lab: every code; shipped build: the ones pick_h2p_tile returns)
//   0: 128 x 128, 4 waves of 64 x 64, 3 waves / SIMD     1: 128 x 128, 2 waves of 64 x 128     2: 128 x 128, 2 waves of 128 x 64
//   3: 128 x 256, 4 waves of 64 x 128                    4: 128 x 256, 4 waves of 128 x 64     5: 128 x 512, 4 waves of 128 x 128 (1 wave / SIMD)
//   6: 128 x 256, 2 waves of 128 x 128                   7: 128 x 128, ONE wave of 128 x 128   8: 64 x 128 per one-wave workgroup
//   9: 64 x 64 per one-wave workgroup                   10: 128 x 64 per one-wave workgroup
#ifndef H2P_TILES
#define H2P_TILES 0x7ff
#endif
static int pick_h2p_tile(const H2PGroup& G) {
    // 128 x 128 tiles of the launch; from ~3 per CU on the 128 x 256 workgroup of four 64 x 128 waves (two waves per SIMD: measured 0.77-0.92 of the
    // matrix pipe inside the K loop against 0.70-0.83 for three 64 x 64 waves, profiles/r06_kloop_h2p_trace.log), below that the finer 64 x 64 grain
    const long n128 = (long)((G.max_n + 127) / 128) * (G.g[0].Cout / 128) * G.B * G.n;
    return n128 >= 768 ? 3 : 0;
}

void conv_h2p_group(const H2PGroup& Gin, hipStream_t st, int tile) {
    H2PGroup G = Gin;
    if (G.max_n <= 0 || G.B <= 0) return;
    for (int i = 1; i < G.n; i++)                       // longest K loop first
        for (int j = i; j > 0 && (long)G.g[j].ntap * G.g[j].Cin > (long)G.g[j - 1].ntap * G.g[j - 1].Cin; j--) {
            H2PArgs t = G.g[j]; G.g[j] = G.g[j - 1]; G.g[j - 1] = t;
        }
    if (tile < 0 || tile > 10 || !((H2P_TILES >> tile) & 1)) tile = pick_h2p_tile(G);
    switch (tile) {
#if (H2P_TILES >> 1) & 1
        case 1: launch_h2p<2, 4, 2, 1, 2>(G, st); break;
#endif
#if (H2P_TILES >> 2) & 1
        case 2: launch_h2p<4, 2, 1, 2, 2>(G, st); break;
#endif
#if (H2P_TILES >> 3) & 1
        case 3: launch_h2p<2, 4, 2, 2, 2>(G, st); break;
#endif
#if (H2P_TILES >> 4) & 1
        case 4: launch_h2p<4, 2, 1, 4, 2>(G, st); break;
#endif
#if (H2P_TILES >> 5) & 1
        case 5: launch_h2p<4, 4, 1, 4, 1>(G, st); break;
#endif
#if (H2P_TILES >> 6) & 1
        case 6: launch_h2p<4, 4, 1, 2, 1>(G, st); break;
#endif
#if (H2P_TILES >> 7) & 1
        case 7: launch_h2p<4, 4, 1, 1, 1>(G, st); break;
#endif
#if (H2P_TILES >> 8) & 1
        case 8: launch_h2p<2, 4, 1, 1, 2>(G, st); break;
#endif
#if (H2P_TILES >> 9) & 1
        case 9: launch_h2p<2, 2, 1, 1, 3>(G, st); break;
#endif
#if (H2P_TILES >> 10) & 1
        case 10: launch_h2p<4, 2, 1, 1, 2>(G, st); break;
#endif
        default: launch_h2p<2, 2, 2, 2, 3>(G, st); break;
    }
}

// ------------------------------------------------------------------------------------------------
// fp32 [C][ld] -> planes of lrelu(x, slope) + x16 copy: the entry of a stage (the upsampler still writes the channel-major tensor).
// A wave converts 32 positions x 16 channels per step: lane (position, half) reads its 8 channels (two full 128-byte lines per load
// instruction), writes one 16-byte unit per plane and one 32-byte unit of the x16 copy.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void split_planes_kernel(const float* x, long x_ld, int C, long n, float slope, unsigned char* planes, float* x16, long out_ld, unsigned* ovf) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const long pos = ((long)blockIdx.x * 4 + wave) * 32 + l31;
    const int c = blockIdx.y;
    if (pos >= n) return;
    const size_t ps = (size_t)C * (size_t)out_ld * 2u;
    float v[8], t[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = x[(size_t)(16 * c + 8 * (e >> 2) + 4 * half + (e & 3)) * x_ld + pos];
#pragma unroll
    for (int e = 0; e < 8; e++) t[e] = __builtin_fmaxf(v[e], v[e] * slope);
    float amax = 0.f;
    u32x4 ph, pl;
    split8h(t, ph, pl, amax);
    unsigned char* d = planes + (((size_t)c * out_ld + pos) * 32 + half * 16);
    *(u32x4*)d = ph;
    *(u32x4*)(d + ps) = pl;
    if (x16) {
        float* o = x16 + (((size_t)c * out_ld + pos) * 16 + half * 8);
        *(f32x4u*)o = f32x4u{v[0], v[1], v[2], v[3]};
        *(f32x4u*)(o + 4) = f32x4u{v[4], v[5], v[6], v[7]};
    }
    if (amax > kH2Limit && ovf) *ovf = 1u;
}

void split_planes(const float* x, long x_ld, int C, long n, float slope, void* planes, float* x16, long out_ld, unsigned* ovf, hipStream_t st) {
    if (n <= 0 || C <= 0) return;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)((n + 127) / 128), (unsigned)(C / 16)), dim3(256), 0, st, x, x_ld, C, n, slope, (unsigned char*)planes, x16, out_ld, ovf);
}

}  // namespace sts
