// conv_h2w.hip -- the wide ResBlock convs in the WINOGRAD domain on the fp16 matrix cores (round 6, lab: engaged by sts_debug_set only).
//
// Why: the trunk is limited by matrix-core ENERGY at batch (power cap at ~60 % of the nominal MFMA rate) -- a build that simply drops one of
// the three fp16 products per fp32 product (wrong results) runs the matrix-core region 17 / 19 / 22 % faster at 1 / 32 / 64 utterances
// (profiles/r06_h2p_thresholds_and_two_products.log).  The legitimate way to a third fewer products is the segmented F(2,3) / F(2,2) form the
// exact-fp32 path already uses (conv.hip resblock_wino_kernel; /root/reference/src/nn_op/nn_conv1d.cpp:118-199 is the direct form it replaces):
// a k-tap filter = 3-tap segments + 2-tap segments; per segment and PAIR of outputs y(n), y(n + d) of a dilation-d conv
//     d_i = lrelu(x(n + (t0 + i) d - pad)), i = 0..3
//     V0 = d0 - d2   V1 = d1 + d2   V2 = d2 - d1   V3 = d1 - d3            (input transform, fp32, one rounding each)
//     U0 = g0   U1 = (g0 + g1 + g2) / 2   U2 = (g0 - g1 + g2) / 2   U3 = g2   (weight transform, double at load time; 2-tap segment: g2 = 0, no U3)
//     M_j += U_j V_j  over channels and segments                              (4 / 3 matrix products instead of 6 / 4)
//     y(n) = M0 + M1 + M2      y(n + d) = M1 - M2 - M3                        (output transform, once per tile)
// tools/wino_f16x2_numerics.py (pinned by tests/test_split_numerics_cpu.py) models exactly this arithmetic with the two-term fp16 split of the
// transformed operands: relative rms error 1.5-2.1e-7 against float64, BELOW the direct two-term form's.
//
// How (on the pre-split path's infrastructure, conv_h2p.hip): activations travel as x16 (fp32, channel-minor: kernels.hpp H2PArgs) only;
// a workgroup's window of a 16-channel chunk is copied raw into LDS by `buffer_load_dwordx4 ... lds`; a lane of pair column p reads its
// four positions' 8 channels, applies the leaky relu, forms V_j and splits it IN REGISTERS (the B fragment of the next MFMAs: used by the
// wave's 2 row tiles x 3 products and dropped), the U_j fragments come pre-split in fragment order like every other weight.  A wave owns
// 64 rows x 32 pairs (64 outputs): four transform-domain accumulator sets = 128 registers, two waves per SIMD, so one wave's transform
// arithmetic (128-190 VALU per 24 MFMAs) issues under the other's matrix work.
#include "conv_bf3_dev.hpp"

namespace sts {

#define STS_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#ifndef H2W_EXP
#define H2W_EXP 0       // lab (WRONG results): 1 = weight fragments requested for the first chunk only, 2 = input transform + split for the first chunk only, 4 = no LDS reads after the first chunk
#endif

constexpr int W_MAXWIN = 128 + MAX_HALO;       // window positions per workgroup at most: WN (2) x 64 outputs + halo

// two-term split of 8 fp32 values WITHOUT the range tracking of split8h (the producer of an x16 tensor bounds |x|, |V| <= 2 max |x|), the small
// term formed by the mixed-precision FMA: lo' = fp16(fma(-hi, 2048, 2048 v)) = fp16((v - hi) 2048) exactly as split8h (v - hi is exact in fp32);
// 4 VALU per value pair instead of 6 (+ 2 of range tracking): the Winograd form is VALU-bound, not MFMA-bound, with the plain split
__device__ __forceinline__ void split8w(const float (&x)[8], u32x4& hi, u32x4& lo) {
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const f32x2 v = {x[2 * d], x[2 * d + 1]};
        const f16x2 h = __builtin_convertvector(v, f16x2);
        const f32x2 v2 = v * 2048.f;
        const unsigned hu = __builtin_bit_cast(unsigned, h);
        unsigned l = 0u;
        asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
            "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "+v"(l) : "v"(hu), "s"(-2048.0f), "v"(v2[0]), "v"(v2[1]));
        hi[d] = hu;
        lo[d] = l;
    }
}

template <int WM, int WN, int K>
__device__ __forceinline__ void conv_h2w_body(const H2WArgs& a, const SegView& seg, unsigned* const ovf, const int bx, const int by, const int b) {
    constexpr int MW = 2, NWAVE = WM * WN;
    constexpr int RB = 64;                             // bytes of one position of a 16-channel chunk (fp32)
    constexpr int BUFB = W_MAXWIN * RB;                // one window buffer
    constexpr int NSLOT = W_MAXWIN / 16;               // LDS-DMA slots of 16 positions = 1 KB
    constexpr int SPW = (NSLOT + NWAVE - 1) / NWAVE;
    constexpr unsigned ABLK = 2048u;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem3[];
    const int len = uni(seg_len(seg, b));
    const int dil = uni(a.dil), C = uni(a.C);
    constexpr int k = K;
    const int P = dil * (32 / dil);                    // pairs per wave: whole blocks of 2 dil outputs
    const int NTo = WN * 2 * P;                        // outputs per workgroup
    const int n0 = bx * NTo;
    if (n0 >= len) return;
    const size_t base = (size_t)uni(seg_start(seg, b));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int swave = __builtin_amdgcn_readfirstlane(wave);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, half = lane >> 5;
    const int pad = dil * (k - 1) / 2;
    const int Wwin = NTo + (k - 1) * dil;              // window positions in use
    const int win0 = n0 - pad;
    const int mbase = by * (64 * WM) + wm * 64;
    constexpr int ntapw = 4 * (K % 3 == 0 ? K / 3 : (K % 3 == 2 ? (K - 2) / 3 : (K - 4) / 3)) + 3 * (K % 3 == 0 ? 0 : (K % 3 == 2 ? 1 : 2));   // pseudo-taps (U matrices) per chunk
    const int nchunk = C / CK, nrt = C / 32;
    const int nslot = (Wwin + 15) >> 4;

    // pair column of this lane: block of 2 dil outputs, offset inside the block's first half
    const int pp = l31 < P ? l31 : P - 1;              // (idle columns compute a copy of the last pair; never stored)
    const int pcol = wn * 2 * P + (pp / dil) * 2 * dil + (pp % dil);      // output index of y(n) inside the workgroup tile; partner: + dil

    f32x16 acc[4][MW][1];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int i = 0; i < MW; i++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[j][i][0][r] = 0.f;

    // ---- A fragments: pseudo-tap step s = chunk * ntapw + (segment, point), one block of nrt * 2 KB each (bf3_pack perm_k math 1)
    const int nsteps = nchunk * ntapw;
    const rsrc_t wrs = make_rsrc(a.wu, (unsigned)((size_t)nsteps * nrt * ABLK));
    const unsigned a_voff = (unsigned)lane * 16u + (unsigned)(mbase >> 5) * ABLK;
    const unsigned a_step = (unsigned)nrt * ABLK;
    auto load_a = [&](int s, u32x4 (&dst)[MW][2]) {
        const unsigned sb = (unsigned)s * a_step;
#pragma unroll
        for (int i = 0; i < MW; i++)
#pragma unroll
            for (int pl = 0; pl < 2; pl++)
                dst[i][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)a_voff, (int)(sb + (unsigned)i * ABLK + (unsigned)(pl * 1024)), 0));
    };
    // ---- window staging: slot = 16 positions x 64 B; lane l lands at (position l / 4, 16-byte unit l % 4).  A reader of position w finds
    // logical unit u at slot u ^ ((w >> 1) & 3) (pair columns are 2 positions apart at dilation 1: without the rotation 8 lanes of a
    // ds_read_b128 group would share 4 banks), so the lane FETCHES the logical unit that belongs there.  Outside [0, len): zeros.
    unsigned dvoff[SPW];
#pragma unroll
    for (int i = 0; i < SPW; i++) {
        const int sl = swave + i * NWAVE;
        const int w = sl * 16 + (lane >> 2);
        const int pos = win0 + w;
        const bool v = pos >= 0 && pos < len;
        dvoff[i] = v ? (unsigned)pos * 64u + (unsigned)(((lane & 3) ^ ((w >> 1) & 3)) << 4) : kOOB;
    }
    const unsigned char* const xp0 = (const unsigned char*)uni((const void*)a.x16) + base * 64u;
    const size_t xrow = (size_t)uni(a.x_ld) * 64u;
    const unsigned span = (unsigned)len * 64u;
    auto dma = [&](int c, int bufi) {
        const rsrc_t rs = make_rsrc(xp0 + (size_t)c * xrow, span);
#pragma unroll
        for (int i = 0; i < SPW; i++) {
            const int sl = swave + i * NWAVE;
            if (sl < nslot) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, STS_LDS_PTR(smem3 + bufi * BUFB + sl * 1024), 16, (int)dvoff[i], 0, 0, 0);
        }
    };
    // ---- the four inputs d_i of this lane's pair for segment (first tap t0): 8 channels (unit pair 2 half, 2 half + 1) of 4 positions
    const float slope = a.in_slope;
    const bool act = slope != 1.0f;                 // (uniform; the second conv of a layer reads an already activated tensor)
    float amax = 0.f;
    auto load_d = [&](const int bufi, const int t0, const bool three, float (&d)[4][8]) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i == 3 && !three) { for (int e = 0; e < 8; e++) d[3][e] = 0.f; continue; }       // (a 2-tap segment has no point 3: its d3 would lie past the window)
            const int w = pcol + (t0 + i) * dil;
            const unsigned char* row = smem3 + bufi * BUFB + w * RB;
            const int rot = (w >> 1) & 3;
            const f32x4u lo4 = *(const f32x4u*)(row + (((2 * half) ^ rot) << 4));
            const f32x4u hi4 = *(const f32x4u*)(row + (((2 * half + 1) ^ rot) << 4));
#pragma unroll
            for (int e = 0; e < 4; e++) { d[i][e] = act ? __builtin_fmaxf(lo4[e], lo4[e] * slope) : lo4[e]; d[i][4 + e] = act ? __builtin_fmaxf(hi4[e], hi4[e] * slope) : hi4[e]; }
        }
    };
    // V_j of the pair, split into its two fp16 terms = the B fragment of point j
    auto make_v = [&](const float (&d)[4][8], int j, u32x4 (&v)[1][2]) {
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; e++)
            t[e] = j == 0 ? d[0][e] - d[2][e] : (j == 1 ? d[1][e] + d[2][e] : (j == 2 ? d[2][e] - d[1][e] : d[1][e] - d[3][e]));
        split8w(t, v[0][0], v[0][1]);
    };

    // ---- main loop: chunks (window double-buffered through LDS-DMA, one barrier each) x segments x points.  Step s reads fa[s & 1] and requests
    // step s + 1 into the other buffer; buffer indices must be compile-time constants (register arrays) and a chunk has an odd number of steps
    // for k = 5 / 11, so the loop body is TWO chunks with every (segment, point) unrolled -- the kernel is instantiated per tap count K.
    constexpr int N3 = K % 3 == 0 ? K / 3 : (K % 3 == 2 ? (K - 2) / 3 : (K - 4) / 3), N2 = K % 3 == 0 ? 0 : (K % 3 == 2 ? 1 : 2);
    constexpr int NSEG = N3 + N2, NTAPW = 4 * N3 + 3 * N2;
    dma(0, 0);
    if (nchunk > 1) dma(1, 1);
    u32x4 fa[2][MW][2];
    load_a(0, fa[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c0 = 0; c0 < nchunk; c0 += 2) {
        static_for<0, 2>([&](auto ccc) {
            constexpr int cc = decltype(ccc)::value;
            const int c = c0 + cc;
            if (c < nchunk) {
                static_for<0, NSEG>([&](auto sgc) {
                    constexpr int sg = decltype(sgc)::value;
                    constexpr bool three = sg < N3;
                    constexpr int t0 = three ? 3 * sg : 3 * N3 + 2 * (sg - N3);
                    constexpr int q0 = three ? 4 * sg : 4 * N3 + 3 * (sg - N3);          // first pseudo-tap of the segment
                    float d[4][8];
                    if (!(H2W_EXP & 4) || c == 0) load_d(cc, t0, three, d); else { for (int i = 0; i < 4; i++) for (int e = 0; e < 8; e++) d[i][e] = (float)(i + e); }
                    static_for<0, (three ? 4 : 3)>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        constexpr int par = (cc * NTAPW + q0 + j) & 1;
                        u32x4 vb[1][2];
                        if (!(H2W_EXP & 1) || c == 0) load_a(c * NTAPW + q0 + j + 1, fa[par ^ 1]);                    // past the last step: zeros beyond the descriptor
                        if (!(H2W_EXP & 2) || c == 0) make_v(d, j, vb); else { vb[0][0] = fa[par][0][0]; vb[0][1] = fa[par][0][1]; }
                        step_mfmas<1, MW, 1, 2, 2>(acc[j], fa[par], vb);
                    });
                });
                if (c + 1 < nchunk) {
                    // my slots of chunk c + 1 have landed: they were requested a chunk ago, loads complete in order, and at most the 2 MW fragment
                    // loads of the step just prefetched are younger -- waiting for all but those does not expose their latency
                    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    __syncthreads();                                        // everyone's have; everyone is done reading chunk c's buffer
                    if (c + 2 < nchunk) dma(c + 2, cc);
                }
            }
        });
    }

    // ---- output transform + epilogue.  Register hh * 8 + e of a 32-row tile i = channel 16 (mbase / 16 + 2 i + hh) + 8 (e >> 2) + 4 half + (e & 3).
    if (l31 >= P) { if (amax > kH2Limit && ovf) *ovf = 1u; return; }
    const float ws = a.wscale;
    const float* const bias = uni(a.bias);
    const float* const res16 = uni(a.res16);
    float* const y16 = uni(a.y16);
    float* const y = uni(a.y);
    const float oslope = a.out_slope;
    const int posA = n0 + pcol, posB = posA + dil;
    static_for<0, MW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int cc0 = (mbase >> 4) + 2 * i;
#pragma unroll
        for (int o = 0; o < 2; o++) {                   // o = 0: y(n), 1: y(n + d)
            const int pos = o ? posB : posA;
            if (pos >= len) continue;
#pragma unroll
            for (int hh = 0; hh < 2; hh++) {
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; e++) {
                    const int r = hh * 8 + e;
                    const float m = o == 0 ? (acc[0][i][0][r] + acc[1][i][0][r]) + acc[2][i][0][r] : (acc[1][i][0][r] - acc[2][i][0][r]) - acc[3][i][0][r];
                    const int ch = 16 * (cc0 + hh) + 8 * (e >> 2) + 4 * half + (e & 3);
                    v[e] = m * ws + (bias ? bias[ch] : 0.f);
                }
                if (res16) {
                    const float* rq = res16 + ((size_t)(cc0 + hh) * a.res_ld + base + pos) * 16 + half * 8;
                    const f32x4u r0 = *(const f32x4u*)rq, r1 = *(const f32x4u*)(rq + 4);
#pragma unroll
                    for (int e = 0; e < 4; e++) { v[e] += r0[e]; v[4 + e] += r1[e]; }
                }
                if (y) {
#pragma unroll
                    for (int e = 0; e < 8; e++) y[(size_t)(16 * (cc0 + hh) + 8 * (e >> 2) + 4 * half + (e & 3)) * a.y_ld + base + pos] = v[e];
                }
                if (y16) {
                    float* q = y16 + ((size_t)(cc0 + hh) * a.y16_ld + base + pos) * 16 + half * 8;
                    *(f32x4u*)q = f32x4u{__builtin_fmaxf(v[0], v[0] * oslope), __builtin_fmaxf(v[1], v[1] * oslope), __builtin_fmaxf(v[2], v[2] * oslope), __builtin_fmaxf(v[3], v[3] * oslope)};
                    *(f32x4u*)(q + 4) = f32x4u{__builtin_fmaxf(v[4], v[4] * oslope), __builtin_fmaxf(v[5], v[5] * oslope), __builtin_fmaxf(v[6], v[6] * oslope), __builtin_fmaxf(v[7], v[7] * oslope)};
                }
            }
        }
    });
    if (amax > kH2Limit && ovf) *ovf = 1u;
}

template <int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_h2w_group_kernel(H2WGroup G, int mtiles, int nx) {
    const TileId t = map_tile(nx, mtiles, G.B * G.n);
    if (!t.valid) return;
    const int gi = t.bz / G.B, b = t.bz - gi * G.B;
    const H2WArgs* ga = (const H2WArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const int k = uni(ga[gi].k);        // (the members of a grouped launch differ in their tap count: one unrolled body per count)
    if (k == 3) conv_h2w_body<WM, WN, 3>(ga[gi], G.seg, G.ovf, t.bx, t.by, b);
    else if (k == 7) conv_h2w_body<WM, WN, 7>(ga[gi], G.seg, G.ovf, t.bx, t.by, b);
    else if (k == 11) conv_h2w_body<WM, WN, 11>(ga[gi], G.seg, G.ovf, t.bx, t.by, b);
    else conv_h2w_body<WM, WN, 5>(ga[gi], G.seg, G.ovf, t.bx, t.by, b);
}

bool conv_h2w_group_eligible(const H2WGroup& G) {
    if (G.n < 1 || G.n > kMaxGroup || G.max_n <= 0 || G.B <= 0) return false;
    for (int i = 0; i < G.n; i++) {
        const H2WArgs& a = G.g[i];
        if (!a.x16 || !a.wu || (!a.y && !a.y16)) return false;
        if (a.C % 128 != 0 || a.C != G.g[0].C || a.dil != G.g[0].dil) return false;      // (one pair geometry per launch: members share the dilation)
        if ((a.k != 3 && a.k != 5 && a.k != 7 && a.k != 11) || a.dil < 1 || a.dil > 16 || (a.k - 1) * a.dil > MAX_HALO) return false;
        if ((double)a.x_ld * 64.0 >= 2.0e9) return false;
    }
    return true;
}

void conv_h2w_group(const H2WGroup& Gin, hipStream_t st) {
    H2WGroup G = Gin;
    if (G.max_n <= 0 || G.B <= 0) return;
    for (int i = 1; i < G.n; i++)                       // longest K loop first
        for (int j = i; j > 0 && G.g[j].k > G.g[j - 1].k; j--) { H2WArgs t = G.g[j]; G.g[j] = G.g[j - 1]; G.g[j - 1] = t; }
    const int dil = G.g[0].dil, P = dil * (32 / dil), NTo = 2 * 2 * P;
    const int mt = G.g[0].C / 128;
    const int nx = (G.max_n + NTo - 1) / NTo;
    constexpr size_t lds = (size_t)2 * W_MAXWIN * 64;
    hipLaunchKernelGGL((conv_h2w_group_kernel<2, 2>), dim3(mapped_grid(nx, mt, G.B * G.n)), dim3(256), lds, st, G, mt, nx);
}

// U pseudo-taps of a conv w[k][Cin][Cout] (the engine's packed fp32 order) -> [ntapw][Cin][Cout] fp32 (double arithmetic, rounded once),
// in the order the kernel walks them: segment by segment, points 0..3 (0..2 for a 2-tap segment).  Returns ntapw.
int h2w_transform_weights(const float* wp, int k, int Cin, int Cout, float* dst) {
    int n3, n2; wino_split(k, &n3, &n2);
    const size_t slab = (size_t)Cin * Cout;
    int q = 0;
    for (int sg = 0; sg < n3 + n2; sg++) {
        const int t0 = sg < n3 ? 3 * sg : 3 * n3 + 2 * (sg - n3), tl = sg < n3 ? 3 : 2;
        for (int j = 0; j < (tl == 3 ? 4 : 3); j++, q++)
            for (size_t i = 0; i < slab; i++) {
                const double g0 = wp[(size_t)t0 * slab + i], g1 = wp[(size_t)(t0 + 1) * slab + i], g2 = tl == 3 ? (double)wp[(size_t)(t0 + 2) * slab + i] : 0.0;
                const double u = j == 0 ? g0 : (j == 1 ? 0.5 * (g0 + g1 + g2) : (j == 2 ? 0.5 * (g0 - g1 + g2) : g2));
                dst[(size_t)q * slab + i] = (float)u;
            }
    }
    return q;
}

// fp32 [C][ld] -> x16 (channel-minor fp32 copy): the entry of a stage on this path
__global__ __launch_bounds__(256) void to_x16_kernel(const float* x, long x_ld, long n, float* x16, long out_ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const long pos = ((long)blockIdx.x * 4 + wave) * 32 + l31;
    const int c = blockIdx.y;
    if (pos >= n) return;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = x[(size_t)(16 * c + 8 * (e >> 2) + 4 * half + (e & 3)) * x_ld + pos];
    float* o = x16 + (((size_t)c * out_ld + pos) * 16 + half * 8);
    *(f32x4u*)o = f32x4u{v[0], v[1], v[2], v[3]};
    *(f32x4u*)(o + 4) = f32x4u{v[4], v[5], v[6], v[7]};
}
void to_x16(const float* x, long x_ld, int C, long n, float* x16, long out_ld, hipStream_t st) {
    if (n <= 0 || C <= 0) return;
    hipLaunchKernelGGL(to_x16_kernel, dim3((unsigned)((n + 127) / 128), (unsigned)(C / 16)), dim3(256), 0, st, x, x_ld, n, x16, out_ld);
}

}  // namespace sts
