// wn_flow.hip -- the reverse normalising flow as ONE launch per WaveNet layer (round 4).
//
// Reference: /root/reference/src/modules/WN.cpp:100-149 (in_layer conv -> tanh * sigmoid -> res_skip 1x1 -> x += res, output += skip),
// ResidualCouplingLayer.cpp:47-66 (h = pre(x0); m = post(WN(h)); x1 -= m), ResidualCouplingBlock.cpp:59-70 (couplings in reverse order).
//
// Why.  At one utterance the flow was 40 dependent launches of 9-14 us (pre, 4 x (gate conv, res/skip conv), post per coupling), each a
// chip-wide launch whose price is the launch itself (docs/HISTORY.md 5b), on the exact-fp32 MFMA with dword operand loads.  Fusing the gate conv
// with the res/skip conv that consumes it needs a workgroup that owns ALL 2H gate rows of its columns -- 21 workgroups for 668 frames,
// compute-starved (docs/HISTORY.md 5b) -- unless the 1x1 conv is cut along K instead:
//
//   * the H gated channels are cut into G groups of Cg = 32 (16) channels.  Workgroup (column tile of 32 frames, group g) stages the
//     layer's whole input window h (all H channels x 32 + 2 halo frames), runs the gate conv for ITS 2 Cg gate rows only (K = H x k),
//     applies tanh * sigmoid in registers, parks the Cg gated channels in LDS and multiplies them with the matching K slice of the
//     res/skip matrix: a PARTIAL sum of all res (and skip) rows.  Partial sums leave through memory; the next layer's staging adds the G
//     partials to h in a fixed order (bit-reproducible: no atomics) -- G + 1 reads of a 27 KB window per workgroup instead of one, which
//     is what balances the weight stream (per workgroup: 245 KB of gate weights at G = 6 against 491 KB at G = 3).
//   * skip never materialises: post is linear, so post(sum_l skip_l) = sum_l (post . skip_l) -- the composite (half x H) matrices
//     -(W_post W_skip_l) are formed in double at load time (model.hip flow_pack) and ride as extra rows of each layer's 1x1 conv.  Every
//     workgroup accumulates its partial of -m over the coupling's layers in a private slice (macc[g], its own columns: no races); the
//     NEXT coupling's first layer adds the G slices to the half it reads as x0 (that half is this coupling's x1), writes it back to z and
//     feeds it to its own `pre` conv, which runs inside the same launch (K = C/2: 0.5 us); after the last coupling a small kernel does it.
//   * all matrix work on v_mfma_f32_32x32x16_f16 with two-term operands (conv_bf3_dev.hpp: 3 products per fp32 product, fp32
//     accumulation), operands through 16-byte loads: the intermediate tensors (h, partials, macc) are CHANNEL-MINOR [frame][channel],
//     which makes a lane's 4 accumulator rows one 16-byte store and the staged window a run of contiguous 768-byte rows.
//   Launches per coupling: n_layers (4) instead of 2 n_layers + 2 (10); per step 17 instead of 40.
#include "conv_bf3_dev.hpp"

namespace sts {

#ifdef STS_TILE_TRACE
int tile_trace_bind_flow(long long* buf, unsigned capacity_records) { return tile_trace_bind(buf, capacity_records); }
#endif

constexpr int FL_WIN = 64;            // staged window: 32 output frames + 2 * halo <= 64 positions
constexpr int FL_NT = 32;             // output frames per workgroup
constexpr int FL_UR = 2;              // staging: (quad of channels, position) units per thread and round (each with up to 9 16-byte loads in flight)
constexpr int FL_MAXG = 8;            // channel groups (partial-sum slices) at most

__device__ __forceinline__ unsigned fl_b_off(int chunk, int pos, int half, int plane, int plane_bytes) {
    // B-operand planes in LDS: [chunk of 16 channels][plane hi / lo'][position][2 x 8 channels], the 16-byte half XOR-swizzled with bit 3
    // of the position (conflict-free ds_read_b128 for every tap shift, as in conv_bf3_body)
    return (unsigned)(chunk * 2 * plane_bytes + plane * plane_bytes + pos * 32 + ((half ^ ((pos >> 3) & 1)) << 4));
}

// split 4 consecutive channels (one fp32 quad) into their two fp16 terms: 8 bytes per plane
__device__ __forceinline__ void split4h(const float (&x)[4], unsigned (&hi)[2], unsigned (&lo)[2], float& amax) {
#pragma unroll
    for (int d = 0; d < 2; d++) {
        const f32x2 v = {x[2 * d], x[2 * d + 1]};
        const f16x2 h = __builtin_convertvector(v, f16x2);
#if STS_SPLIT_MIX        // (conv_bf3_dev.hpp split8h: the small term as one mixed-precision FMA per value, same bits)
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

// 8 waves per workgroup.  Everything that does not depend on the previous layer -- this wave's share of the gate conv's weights (all of it:
// <= FL_GSTEPS steps of 2 KB), the 1x1 conv's weights -- is requested up front and lands while the window is staged, so the K loop
// itself reads LDS only.
constexpr int FL_WAVES = 8;
constexpr int FL_GSTEPS = 16;         // gate-conv steps (16 channels x 1 tap) a wave may own: H k / (16 KS) <= 16
template <bool L0>        // L0: first layer of a coupling (x0 staging + `pre` inside the launch) -- separate kernels, separate register budgets
__global__ __launch_bounds__(64 * FL_WAVES) __attribute__((amdgpu_waves_per_eu(2, 2))) void flow_layer_kernel(FlowLayerArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
    const int Gp = (a.G + 7) & ~7;
    const int g = (int)(blockIdx.x % (unsigned)Gp);
    if (g >= a.G) return;
    const int rest = (int)(blockIdx.x / (unsigned)Gp);
    const int ntile = (a.max_len + FL_NT - 1) / FL_NT;
    const int b = rest / ntile, tile = rest - b * ntile;
    const int len = uni(seg_len(a.seg, b));
    const int n0 = tile * FL_NT;
    if (n0 >= len) return;
    const size_t base = (size_t)uni(seg_start(a.seg, b));
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int H = a.H, hc = a.half, Cg = a.Cg, halo = a.halo;
    const int W = FL_NT + 2 * halo;                       // window positions in use
    const int NCH = H >> 4;                               // 16-channel chunks of h
    const int PLANE = FL_WIN * 32;                        // bytes of one plane of one chunk
    unsigned char* const R_h = fsm;                                         // h planes: NCH chunks
    unsigned char* const R_x = fsm + (size_t)NCH * 2 * PLANE;              // layer 0: x0 planes (hc / 16 chunks); later: K-split exchange
    unsigned char* const R_a = R_x + (size_t)FL_WAVES * 4096;              // gated planes: (Cg / 16) chunks x 2 planes x 32 positions x 32 B
    const int APL = FL_NT * 32;
    float amax = 0.f;
    constexpr int RG = 2, KS = FL_WAVES / RG;             // Cg = 32: two gate row tiles per group, four K slices (waves) per row tile
    const int rt_l = wave % RG, kq = wave / RG;
#ifdef STS_TILE_TRACE
    long long* tt_rec = tt_open(3 + (L0 ? 1 : 0), g);    // kind 3 / 4 = flow layer (stamps: start, window staged, operands in LDS, gate K loop done, gated planes parked, 1x1 done + stores)
    TT_STAMP(0);
#endif

    // ---- gate conv: this wave's (row tile, K slice); its whole weight stream is requested now
    const int nrtg = (2 * H) >> 5;
    const int c_lo = NCH * kq / KS, c_hi = NCH * (kq + 1) / KS;
    const int nsteps = (c_hi - c_lo) * a.k;               // <= FL_GSTEPS (flow_layer_shape_ok)
    u32x4 fa[FL_GSTEPS][1][2];
    auto gate_prefetch = [&](auto lo_c, auto hi_c) {      // steps [lo, hi) of this wave
        constexpr int lo = decltype(lo_c)::value, hi = decltype(hi_c)::value;
        const rsrc_t wrs = make_rsrc(a.w_gate, (unsigned)((size_t)NCH * a.k * nrtg * 2048u));
        const int rt = g * RG + rt_l;
#pragma unroll
        for (int s = lo; s < hi; s++) {
            const int c = c_lo + s / a.k, j = s - (s / a.k) * a.k;
            const unsigned o = s < nsteps ? (unsigned)(((c * a.k + j) * nrtg + rt) * 2048 + lane * 16) : kOOB;
#pragma unroll
            for (int pl = 0; pl < 2; pl++) fa[s][0][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)o, pl * 1024, 0));
        }
    };
    using I0 = std::integral_constant<int, 0>; using IH = std::integral_constant<int, FL_GSTEPS / 2>; using IE = std::integral_constant<int, FL_GSTEPS>;
    // ---- the 1x1 conv's weights: <= 2 row tiles per wave x <= 4 K steps
    const int ntc = a.rows_c >> 5;                        // 32-row tiles of the res + m conv
    const int nkc = Cg >> 4;                              // its K steps (this group's gated channels)
    u32x4 fc[2][2][1][2];                                 // (Cg = 32: two K steps)
    auto c_prefetch = [&]() {
        const rsrc_t wrs = make_rsrc(a.w_c, (unsigned)((size_t)NCH * ntc * 2048u));
#pragma unroll
        for (int ti = 0; ti < 2; ti++) {
            const int t = wave + FL_WAVES * ti;
#pragma unroll
            for (int cs = 0; cs < 2; cs++)
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    const unsigned off = t < ntc && cs < nkc ? (unsigned)(((g * nkc + cs) * ntc + t) * 2048 + pl * 1024 + lane * 16) : kOOB;
                    fc[ti][cs][0][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)off, 0, 0));
                }
        }
    };

    // ---- the -m accumulators this wave will update (read-modify-write of its own columns): requested after the staging, needed in the last epilogue
    f32x4u mold[2][4];
    auto m_prefetch = [&]() {
        const rsrc_t rm = make_rsrc(a.macc + (size_t)g * a.macc_stride, a.macc_init ? 0u : (unsigned)(((size_t)a.tot * hc) * 4u));
#pragma unroll
        for (int ti = 0; ti < 2; ti++) {
            const int t = wave + FL_WAVES * ti;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int c = t * 32 + 8 * q + 4 * half - a.rows_res;
                const bool okm = t < ntc && c >= 0 && c + 3 < hc && n0 + l31 < len;
                const unsigned off = okm ? (unsigned)(((base + (size_t)(n0 + l31)) * hc + c) * 4u) : kOOB;
                mold[ti][q] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(rm, (int)off, 0, 0));
            }
        }
    };

    // ================= stage the layer's input window h[all H channels][n0 - halo, n0 + 32 + halo) into LDS, split =================
    if constexpr (!L0) {
        // (half of this wave's gate weights are requested together with the window -- a workgroup pulls ~490 KB through its CU's ~100 GB/s
        // path to L2, so what matters is that the path never idles --, the other half + the 1x1 conv's right behind)
        gate_prefetch(I0{}, IH{});
        // h = h_in + part_0 + part_1 + ... (the previous layer's partial res sums, fixed order); units of 4 channels x 1 position.  Branch-free:
        // every source through a buffer descriptor, positions outside the utterance (and sources past part_n) read as zero
        const int qpp = H >> 2;                           // quads per position
        const int nunit = W * qpp;
        const unsigned span = (unsigned)(((size_t)a.tot * H) * 4u);
        const rsrc_t rh = make_rsrc(a.h_in, span);
        for (int u0 = 0; u0 < nunit; u0 += 64 * FL_WAVES * FL_UR) {
            f32x4u v[FL_UR], p[FL_UR][FL_MAXG];
            int pw[FL_UR], q4[FL_UR]; bool ok[FL_UR];
#pragma unroll
            for (int r = 0; r < FL_UR; r++) {
                const int u = u0 + r * 64 * FL_WAVES + tid;
                pw[r] = u / qpp; q4[r] = u - pw[r] * qpp;
                const int pos = n0 - halo + pw[r];
                ok[r] = u < nunit && pos >= 0 && pos < len;
                const unsigned off = ok[r] ? (unsigned)(((base + (size_t)pos) * H + (size_t)q4[r] * 4) * 4u) : kOOB;
                v[r] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(rh, (int)off, 0, 0));
#pragma unroll
                for (int s = 0; s < FL_MAXG; s++) {
                    const rsrc_t rp = make_rsrc(a.part_in + (size_t)s * a.part_stride, s < a.part_n ? span : 0u);
                    p[r][s] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(rp, (int)off, 0, 0));
                }
            }
#pragma unroll
            for (int r = 0; r < FL_UR; r++) {
                const int u = u0 + r * 64 * FL_WAVES + tid;
                if (u >= nunit) continue;
                f32x4u s4 = v[r];
#pragma unroll
                for (int s = 0; s < FL_MAXG; s++) s4 += p[r][s];
                // this group's channel slice of its own columns is the materialised h of this layer (next layer's base)
                const int ch = q4[r] * 4;
                if (ok[r] && pw[r] >= halo && pw[r] < halo + FL_NT && ch >= g * Cg && ch < (g + 1) * Cg)
                    *(f32x4u*)(a.h_out + (base + (size_t)(n0 - halo + pw[r])) * H + ch) = s4;
                const float x4[4] = {s4[0], s4[1], s4[2], s4[3]};
                unsigned hi[2], lo[2];
                split4h(x4, hi, lo, amax);
                const int chunk = ch >> 4, hf = (ch >> 3) & 1, qd = (ch >> 2) & 1;
                unsigned char* d = R_h + fl_b_off(chunk, pw[r], hf, 0, PLANE) + qd * 8;
                *(uint2*)d = uint2{hi[0], hi[1]};
                *(uint2*)(d + PLANE) = uint2{lo[0], lo[1]};
            }
        }
        TT_STAMP(1);
        gate_prefetch(IH{}, IE{});
        c_prefetch();
        m_prefetch();
    } else {
        // ---- first layer of a coupling: x0' = x0 rows (+ the previous coupling's pending -m slices), h = pre(x0') inside this launch.
        gate_prefetch(I0{}, IH{});     // (half of the gate weights now, half behind the x0 staging: the CU's path to L2 is the kernel's bottleneck)
        // `pre`'s weights: wave w owns row tile w of h (H <= 256: <= 8 row tiles) for both 32-position halves of the window
        const int nrt = H >> 5, nk = hc >> 4;
        u32x4 fp[8][1][2];
        {
            const rsrc_t wrs = make_rsrc(a.w_pre, (unsigned)((size_t)nk * nrt * 2048u));
#pragma unroll
            for (int cs = 0; cs < 8; cs++)
#pragma unroll
                for (int pl = 0; pl < 2; pl++) {
                    const unsigned o = wave < nrt && cs < nk ? (unsigned)((cs * nrt + wave) * 2048 + pl * 1024 + lane * 16) : kOOB;
                    fp[cs][0][pl] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)o, 0, 0));
                }
        }
        const int qpp = hc >> 2;
        const int nunit = W * qpp;
        const unsigned pspan = (unsigned)(((size_t)a.tot * hc) * 4u);
        for (int u0 = 0; u0 < nunit; u0 += 64 * FL_WAVES * FL_UR) {
            float zv[FL_UR][4]; f32x4u p[FL_UR][FL_MAXG];
            int pw[FL_UR], q4[FL_UR]; bool ok[FL_UR];
#pragma unroll
            for (int r = 0; r < FL_UR; r++) {
                const int u = u0 + r * 64 * FL_WAVES + tid;
                q4[r] = u / W; pw[r] = u - q4[r] * W;              // position fastest: the four row loads of a lane group are coalesced
                const int pos = n0 - halo + pw[r];
                ok[r] = u < nunit && pos >= 0 && pos < len;
                const size_t zo = (size_t)(q4[r] * 4) * a.x0_ld + base + (size_t)(ok[r] ? pos : 0);
#pragma unroll
                for (int e = 0; e < 4; e++) zv[r][e] = ok[r] ? a.x0[zo + (size_t)e * a.x0_ld] : 0.f;
                const unsigned off = ok[r] ? (unsigned)(((base + (size_t)pos) * hc + (size_t)q4[r] * 4) * 4u) : kOOB;
#pragma unroll
                for (int s = 0; s < FL_MAXG; s++) {
                    const rsrc_t rp = make_rsrc(a.pend + (size_t)s * a.pend_stride, s < a.pend_n ? pspan : 0u);
                    p[r][s] = __builtin_bit_cast(f32x4u, __builtin_amdgcn_raw_buffer_load_b128(rp, (int)off, 0, 0));
                }
            }
#pragma unroll
            for (int r = 0; r < FL_UR; r++) {
                const int u = u0 + r * 64 * FL_WAVES + tid;
                if (u >= nunit) continue;
                f32x4u s4 = {zv[r][0], zv[r][1], zv[r][2], zv[r][3]};
#pragma unroll
                for (int s = 0; s < FL_MAXG; s++) s4 += p[r][s];
                const int ch = q4[r] * 4;
                // the updated half is written out (to x0_out, NOT over x0: neighbouring workgroups still read x0's halo columns and
                // would add the pending slices twice): this group's share of the rows, own columns only
                if (a.x0_out && ok[r] && pw[r] >= halo && pw[r] < halo + FL_NT && (q4[r] % a.G) == g) {
                    float* zp = a.x0_out + (size_t)ch * a.x0_out_ld + base + (size_t)(n0 - halo + pw[r]);
#pragma unroll
                    for (int e = 0; e < 4; e++) zp[(size_t)e * a.x0_out_ld] = s4[e];
                }
                const float x4[4] = {s4[0], s4[1], s4[2], s4[3]};
                unsigned hi[2], lo[2];
                split4h(x4, hi, lo, amax);
                const int chunk = ch >> 4, hf = (ch >> 3) & 1, qd = (ch >> 2) & 1;
                unsigned char* d = R_x + fl_b_off(chunk, pw[r], hf, 0, PLANE) + qd * 8;
                *(uint2*)d = uint2{hi[0], hi[1]};
                *(uint2*)(d + PLANE) = uint2{lo[0], lo[1]};
            }
        }
        __syncthreads();
        TT_STAMP(1);
        gate_prefetch(IH{}, IE{});     // (lands under the pre conv and the park)
        // h = pre(x0') + b on the whole window
        if (wave < nrt) {
            const int rt = wave;
            f32x16 acc[1][2];
#pragma unroll
            for (int ct = 0; ct < 2; ct++)
#pragma unroll
                for (int r = 0; r < 16; r++) acc[0][ct][r] = 0.f;
#pragma unroll
            for (int cs = 0; cs < 8; cs++)
                if (cs < nk) {
                    u32x4 fb[2][2];
#pragma unroll
                    for (int ct = 0; ct < 2; ct++)
#pragma unroll
                        for (int pl = 0; pl < 2; pl++) fb[ct][pl] = *(const u32x4*)(R_x + fl_b_off(cs, ct * 32 + l31, half, pl, PLANE));
                    step_mfmas<1, 1, 2, 2, 2>(acc, fp[cs], fb);
                }
            // park: + bias, ZERO outside [0, len) (the gate conv's zero padding is padding of h, not of x0), split, in the k order the
            // accumulator layout gives (a lane's rows 4 h + {0..3} and 8 + 4 h + {0..3} of a 16-row block = one 16-byte unit per plane);
            // the first layer's gate weights are packed to match (perm_k)
#pragma unroll
            for (int ct = 0; ct < 2; ct++) {
                const int pw_ = ct * 32 + l31;
                const int pos = n0 - halo + pw_;
                const bool inside = pos >= 0 && pos < len && pw_ < W;
#pragma unroll
                for (int hh = 0; hh < 2; hh++) {
                    float v8[8];
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        const int row = rt * 32 + hh * 16 + 8 * (e >> 2) + 4 * half + (e & 3);
                        const float t1 = acc[0][ct][hh * 8 + e] * a.s_pre + (a.b_pre ? a.b_pre[row] : 0.f);
                        v8[e] = inside ? t1 : 0.f;
                    }
                    u32x4 ph, pl;
                    split8h(v8, ph, pl, amax);
                    unsigned char* d = R_h + fl_b_off(rt * 2 + hh, pw_, half, 0, PLANE);
                    *(u32x4*)d = ph;
                    *(u32x4*)(d + PLANE) = pl;
                    // this group's channel slice of its own columns -> h_out (channel-minor): quads of 4 consecutive channels
                    if (inside && pw_ >= halo && pw_ < halo + FL_NT) {
#pragma unroll
                        for (int qd = 0; qd < 2; qd++) {
                            const int ch = rt * 32 + hh * 16 + 8 * qd + 4 * half;
                            if (ch >= g * Cg && ch < (g + 1) * Cg)
                                *(f32x4u*)(a.h_out + (base + (size_t)pos) * H + ch) = f32x4u{v8[4 * qd], v8[4 * qd + 1], v8[4 * qd + 2], v8[4 * qd + 3]};
                        }
                    }
                }
            }
        }
        c_prefetch();
        m_prefetch();
    }
    __syncthreads();
    TT_STAMP(2);

    // ================= gate conv: this group's 2 Cg rows, K = H x k, K split over KS waves per row tile; operands: registers + LDS =================
    // (two accumulators, even / odd steps: one accumulator tile would make the wave's 48 MFMAs one dependent chain of ~64 cycles each)
    f32x16 acc[1][1], acc1[1][1];
#pragma unroll
    for (int r = 0; r < 16; r++) { acc[0][0][r] = 0.f; acc1[0][0][r] = 0.f; }
    {
        u32x4 fb[2][1][2];
        auto load_b = [&](int s, u32x4 (&dst)[1][2]) {
            const int ss = s < nsteps ? s : 0;
            const int c = c_lo + ss / a.k, j = ss - (ss / a.k) * a.k;
#pragma unroll
            for (int pl = 0; pl < 2; pl++) dst[0][pl] = *(const u32x4*)(R_h + fl_b_off(c, l31 + j, half, pl, PLANE));
        };
        load_b(0, fb[0]);
        static_for<0, FL_GSTEPS>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            if (s < nsteps) {
                load_b(s + 1, fb[(s + 1) % 2]);
                if constexpr (s % 2 == 0) step_mfmas<1, 1, 1, 2, 2>(acc, fa[s], fb[s % 2]);
                else step_mfmas<1, 1, 1, 2, 2>(acc1, fa[s], fb[s % 2]);
            }
        });
#pragma unroll
        for (int r = 0; r < 16; r++) acc[0][0][r] += acc1[0][0][r];
    }
    TT_STAMP(3);
    // ---- combine the K slices (fixed order), gate, park the Cg gated channels as the 1x1 conv's B operand.  Every wave publishes its
    // partial tile; wave (row tile, kq) then finishes the channel pairs r in {2 kq, 2 kq + 1} of that row tile (KS = 4 slices x 2 pairs = the 8
    // (tanh, sigmoid) pairs a lane holds): the exp-based tanh / sigmoid are ~100 VALU instructions per value -- on two waves out of
    // eight they were 1.5 us of the kernel
    float* const red = (float*)R_x;                                      // KS x RG tiles of 16 x 64 floats
    __syncthreads();                   // (layer 0: every wave is done reading the x0 planes this region held)
#pragma unroll
    for (int r = 0; r < 16; r++) red[((size_t)(kq * RG + rt_l) * 16 + r) * 64 + lane] = acc[0][0][r];
    __syncthreads();
    {
        const int rowb = (g * RG + rt_l) * 32;
        float gv[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int r = 2 * kq + i;                   // pair index 0..7 (register r and r + 8)
            float t = 0.f, sg = 0.f;
            for (int q = 0; q < KS; q++) {
                t += red[((size_t)(q * RG + rt_l) * 16 + r) * 64 + lane];
                sg += red[((size_t)(q * RG + rt_l) * 16 + r + 8) * 64 + lane];
            }
            const int rowp = rowb + (r & 3) + 8 * (r >> 2) + 4 * half;
            t = t * a.s_gate + (a.b_gate ? a.b_gate[rowp] : 0.f);
            sg = sg * a.s_gate + (a.b_gate ? a.b_gate[rowp + 16] : 0.f);
            if (a.ubias) { t += a.ubias[(size_t)rowp * a.ubias_ld + b]; sg += a.ubias[(size_t)(rowp + 16) * a.ubias_ld + b]; }
            gv[i] = tanh_ref(t) * sigmoid_ref(sg);
        }
        // two-term split of the pair (as split8h does per dword) -> dword kq of the lane's 16-byte unit in each plane
        const f32x2 v = {gv[0], gv[1]};
        const f16x2 hh = __builtin_convertvector(v, f16x2);
        const f32x2 rr = (v - __builtin_convertvector(hh, f32x2)) * 2048.f;
        const f16x2 ll = __builtin_convertvector(rr, f16x2);
        amax = __builtin_fmaxf(amax, __builtin_fmaxf(__builtin_fabsf(v[0]), __builtin_fabsf(v[1])));
        unsigned char* d = R_a + (size_t)rt_l * 2 * APL + l31 * 32 + ((half ^ ((l31 >> 3) & 1)) << 4) + kq * 4;
        *(unsigned*)d = __builtin_bit_cast(unsigned, hh);
        *(unsigned*)(d + APL) = __builtin_bit_cast(unsigned, ll);
    }
    __syncthreads();
    TT_STAMP(4);

    // ================= 1x1 conv on the gated channels: partial sums of the res rows and of the -m rows =================
    const int pos_o = n0 + l31;
    const bool col_ok = pos_o < len;
#pragma unroll
    for (int ti = 0; ti < 2; ti++) {
        const int t = wave + FL_WAVES * ti;
        if (t >= ntc) continue;
        f32x16 ac[1][1];
#pragma unroll
        for (int r = 0; r < 16; r++) ac[0][0][r] = 0.f;
#pragma unroll
        for (int cs = 0; cs < 2; cs++)
            if (cs < nkc) {
                u32x4 fb[1][2];
                const unsigned char* sp = R_a + (size_t)cs * 2 * APL + l31 * 32 + ((half ^ ((l31 >> 3) & 1)) << 4);
                fb[0][0] = *(const u32x4*)sp;
                fb[0][1] = *(const u32x4*)(sp + APL);
                step_mfmas<1, 1, 1, 2, 2>(ac, fc[ti][cs], fb);
            }
        if (!col_ok) continue;
        const int row0 = t * 32;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = row0 + 8 * q + 4 * half;             // the lane's 4 consecutive rows of this register group
            f32x4u v = {ac[0][0][4 * q] * a.s_c, ac[0][0][4 * q + 1] * a.s_c, ac[0][0][4 * q + 2] * a.s_c, ac[0][0][4 * q + 3] * a.s_c};
            if (row < a.rows_res) {
                // res rows (layers before the last): partial of x += res; the bias rides on group 0
                if (g == 0 && a.b_res) v += *(const f32x4u*)(a.b_res + row);
                if (row + 3 < H) *(f32x4u*)(a.part_out + (size_t)g * a.part_stride + (base + (size_t)pos_o) * H + row) = v;
            } else {
                const int c = row - a.rows_res;
                if (c + 3 < hc) {
                    float* mp = a.macc + (size_t)g * a.macc_stride + (base + (size_t)pos_o) * hc + c;
                    if (a.macc_init) { if (g == 0 && a.b_m) v += *(const f32x4u*)(a.b_m + c); }
                    else v += mold[ti][q];
                    *(f32x4u*)mp = v;
                }
            }
        }
    }
    if (amax > kH2Limit && a.ovf) *a.ovf = 1u;
#ifdef STS_TILE_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    TT_STAMP(5);
    TT_CLOSE();
#endif
}

// after the last coupling: dst[half rows] = src[half rows] + macc_0 + macc_1 + ... (the -m partial sums; n == 0: a plain copy of a half
// whose newest version lives outside z), channel-minor slices -> channel-major rows
__global__ __launch_bounds__(256) void flow_finish_kernel(FlowFinishArgs a) {
    const int b = blockIdx.y;
    const int len = seg_len(a.seg, b);
    const size_t base = (size_t)seg_start(a.seg, b);
    const int idx = blockIdx.x * 256 + threadIdx.x;          // (quad of channels, position), position fastest
    const int qpp = a.half >> 2;
    const int q = idx / a.max_len, pos = idx - q * a.max_len;
    if (q >= qpp || pos >= len) return;
    f32x4u s4;
#pragma unroll
    for (int e = 0; e < 4; e++) s4[e] = a.src[(size_t)(q * 4 + e) * a.src_ld + base + pos];
    for (int s = 0; s < a.n; s++) s4 += *(const f32x4u*)(a.macc + (size_t)s * a.macc_stride + (base + (size_t)pos) * a.half + q * 4);
#pragma unroll
    for (int e = 0; e < 4; e++) a.dst[(size_t)(q * 4 + e) * a.dst_ld + base + pos] = s4[e];
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
bool flow_layer_shape_ok(int H, int half, int k, int dil, int n_layers) {
    if (H < 32 || H % 32 != 0 || H > 256 || half < 16 || half % 16 != 0 || half % 4 != 0 || half > 128) return false;
    if (!(k & 1) || k < 1 || dil != 1 || (k - 1) > FL_WIN - FL_NT || n_layers < 1) return false;
    const int Cg = 32, G = H / Cg;
    if (G > FL_MAXG) return false;
    const int RG = 2, KS = FL_WAVES / RG, NCH = H / 16;
    if (((NCH + KS - 1) / KS) * k > FL_GSTEPS) return false;      // a wave keeps its whole share of the gate weights in registers
    if (half / 16 > 8 || H / 32 > FL_WAVES) return false;          // `pre`: <= 8 K steps, one row tile per wave
    const int rows_c = ((H + half) + 31) / 32 * 32;
    return rows_c / 32 <= 2 * FL_WAVES;     // <= 2 row tiles of the 1x1 conv per wave
}
int flow_layer_groups(int H) { return H / 32; }

static size_t flow_lds_bytes(int H) {
    const size_t h = (size_t)(H / 16) * 2 * FL_WIN * 32;
    return h + (size_t)FL_WAVES * 4096 + (size_t)4 * 2 * FL_NT * 32 + 256;      // + x0 planes / K-split exchange (32 KB) + gated planes
}

void flow_layer(const FlowLayerArgs& a, hipStream_t st) {
    if (a.max_len <= 0 || a.B <= 0) return;
    const int Gp = (a.G + 7) & ~7;
    const int ntile = (a.max_len + FL_NT - 1) / FL_NT;
    const size_t lds = flow_lds_bytes(a.H);
    if (a.layer == 0) {
        (void)hipFuncSetAttribute((const void*)flow_layer_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     // (per device; > 64 KB of dynamic LDS)
        hipLaunchKernelGGL(flow_layer_kernel<true>, dim3((unsigned)((size_t)ntile * a.B * Gp)), dim3(64 * FL_WAVES), lds, st, a);
    } else {
        (void)hipFuncSetAttribute((const void*)flow_layer_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(flow_layer_kernel<false>, dim3((unsigned)((size_t)ntile * a.B * Gp)), dim3(64 * FL_WAVES), lds, st, a);
    }
}

void flow_finish(const FlowFinishArgs& a, hipStream_t st) {
    if (a.max_len <= 0 || a.B <= 0) return;
    const long n = (long)(a.half >> 2) * a.max_len;
    hipLaunchKernelGGL(flow_finish_kernel, dim3((unsigned)((n + 255) / 256), a.B), dim3(256), 0, st, a);
}

}  // namespace sts
