// conv_bf3.hip -- host side of the split-operand convs (weight packing, eligibility, tile choice) and the single-conv launches.
// Device code and the arithmetic: conv_bf3_dev.hpp.  Grouped launches: conv_bf3_group.hip.  Fused ResBlock layer: resblock_bf3.hip.
#include "conv_bf3_dev.hpp"

namespace sts {

#ifdef STS_TILE_TRACE
int tile_trace_bind_group(long long* buf, unsigned capacity_records);       // conv_bf3_group.hip
int tile_trace_bind_resblock(long long* buf, unsigned capacity_records);    // resblock_bf3.hip
int tile_trace_bind_flow(long long* buf, unsigned capacity_records);        // wn_flow.hip
int tile_trace_bind_h2p(long long* buf, unsigned capacity_records);         // conv_h2p.hip
static long long* g_tt_host_buf = nullptr;
extern "C" int sts_debug_tile_trace(long long* buf, unsigned capacity_records) {
    // buf: device memory of (16 + 12 * capacity_records) 64-bit words (null: tracing off)
    g_tt_host_buf = buf;
    if (buf && hipMemset(buf, 0, 16 * sizeof(long long)) != hipSuccess) return -1;
    if (tile_trace_bind(buf, capacity_records) || tile_trace_bind_group(buf, capacity_records) || tile_trace_bind_resblock(buf, capacity_records) || tile_trace_bind_flow(buf, capacity_records) || tile_trace_bind_h2p(buf, capacity_records)) return -1;
    return 0;
}
extern "C" int sts_debug_tile_trace_count() {
    unsigned n = 0;
    if (!g_tt_host_buf || hipMemcpy(&n, g_tt_host_buf, sizeof(unsigned), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return (int)n;
}
#endif

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

bool conv_bf3_eligible(const ConvArgs& a) {
    if (!a.wb3 || a.depthwise || (a.in_reflect && a.transposed)) return false;
    if (a.Cin != a.Cin_pad || a.Cin_pad % CK != 0 || a.Cout_pad % 32 != 0 || a.Cin < 32) return false;
    if ((double)a.x_ld * 64.0 >= 4.0e9) return false;       // 16 rows of a chunk behind one 32-bit buffer descriptor
    const int first = a.tap_off, last = a.tap_off + (a.ntap - 1) * a.tap_step;
    const int halo = first < last ? last - first : first - last;
    if (halo > MAX_HALO) return false;
    if (a.epi == EPI_GATE && !a.gate_perm) return false;
    if (a.epi == EPI_TANH_PCM || a.kslices > 1) return false;
    return true;
}

// H2: this tile is also built for the two-term fp16 arithmetic (the tiles the automatic choice uses plus 24; conv_bf3 /
// conv_bf3_group send a MATH 1 conv to no other.  Round 3's sweep of the rest under MATH 1 -- 32 x 128 per wave, 32-channel
// chunks, 64-row tiles -- found nothing better: profiles/r03_f16x2_tile_sweep.log)
// SUM: this tile is also built with the summed-input staging (ConvArgs::nsum >= 2: the upsamplers' tiles, conv_bf3_takes_sum)
template <int MW, int NW, int WM, int WN, int NSUB, int KG = 1, bool H2 = false, bool SUM = false>
static void launch_bf3(const ConvArgs& a, int nphase, hipStream_t st, int pm = 0) {
    constexpr int MT = 32 * MW * WM, NT = 32 * NW * WN;
    pm = pm && a.transposed && (MW == 1 || a.Cout_pad % (32 * MW) == 0);    // a wave's rows must lie inside one phase
    const int mt = pm ? (a.Cout_pad * nphase + MT - 1) / MT : (a.Cout_pad + MT - 1) / MT;
    const int nx = (a.max_n + NT - 1) / NT, ny = pm ? mt : mt * nphase;
    if constexpr (SUM) {
        if (a.nsum >= 2) {
            if (a.math == 1) {
                const size_t lds = bf3_lds_bytes<MW, NW, WM, WN, NSUB, KG, 1>();
                hipLaunchKernelGGL((conv_bf3_kernel<MW, NW, WM, WN, NSUB, KG, 1, true>), dim3(mapped_grid(nx, ny, a.B)), dim3(WM * WN * KG * 64), lds, st, a, mt, nx, ny, pm);
            } else {
                const size_t lds = bf3_lds_bytes<MW, NW, WM, WN, NSUB, KG>();
                hipLaunchKernelGGL((conv_bf3_kernel<MW, NW, WM, WN, NSUB, KG, 0, true>), dim3(mapped_grid(nx, ny, a.B)), dim3(WM * WN * KG * 64), lds, st, a, mt, nx, ny, pm);
            }
            return;
        }
    }
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

// 20: 128 x 128 with K split over two wave groups inside the workgroup (8 waves, 32-channel staged chunks)
// 24: 256 x 64, K split over two wave groups (8 waves): all rows of a 256-channel conv behind ONE staged window
// (ConvArgs::rowph: Cout_pad counts the merged rows cout * stride + phase; the tile choice keeps reasoning per phase)
static inline int bf3_phases(const ConvArgs& a) { return a.transposed ? a.out_stride : 1; }
static inline int bf3_phase_rows(const ConvArgs& a) { return a.rowph ? a.Cout_pad / a.rowph : a.Cout_pad; }
static inline int bf3_pick(const ConvArgs& a) { return pick_bf3_tile(bf3_phase_rows(a), a.max_n, (long)a.B * bf3_phases(a), a.transposed != 0, a.math); }

long conv_bf3_blocks(const ConvArgs& a) {
    const int nphase = bf3_phases(a);
    const int tile = bf3_pick(a);
    const int mt = tile == 4 ? 32 : (tile == 3 ? 64 : 128), nt = tile == 4 ? 256 : 128;
    return (long)((a.max_n + nt - 1) / nt) * ((bf3_phase_rows(a) + mt - 1) / mt) * nphase * a.B;
}

// the tiles instantiated with the summed-input staging: the ones the automatic choice gives a transposed conv (upsamplers)
bool conv_bf3_takes_sum(const ConvArgs& a) {
    if (!conv_bf3_eligible(a) || a.in_reflect || a.nsum < 2 || a.nsum > 3 || !a.xs1 || (a.nsum == 3 && !a.xs2)) return false;
    const int nphase = bf3_phases(a);
    const int tile = bf3_pick(a);
    // every workgroup that stages a window forms the mean itself (3 reads + a division per staged value), so the fold only pays where a
    // window is staged once or twice: the phase-merged tiles whose row space (phases x Cout) fits one or two workgroups.  Measured
    // (profiles/r04_ab_log.md): stage-2 upsampler, 8 phases x 128 rows on 128-row tiles: 61 us folded vs 48 + 6; stages 3 / 4: 34 vs 25.5 + 14
    if (tile == 22) return (long)nphase * bf3_phase_rows(a) <= 2 * 128;
    if (tile == 23) return (long)nphase * bf3_phase_rows(a) <= 2 * 64;
    return false;
}

void conv_bf3(const ConvArgs& a, hipStream_t st, int tile) {
    const int nphase = a.rowph ? 1 : bf3_phases(a);        // (row-interleaved phases: ONE row space, launched as a plain conv of Cout_pad merged rows)
    if (a.max_n <= 0 || a.B <= 0) return;
    if (!bf3_tile_ok(tile) || (a.math == 1 && !h2_tile(tile))) tile = bf3_pick(a);
    if (tile >= 8 && tile < 16 && a.Cin_pad % 32 != 0) tile -= 8;
    if (a.nsum >= 2 && tile != 22 && tile != 23) tile = bf3_pick(a);
    switch (tile) {
        case 20: if (a.Cin_pad % 32 == 0) { launch_bf3<2, 2, 2, 2, 2, 2, true>(a, nphase, st); break; } launch_bf3<2, 2, 2, 2, 1, 1, true>(a, nphase, st); break;
        // phase-merged rows (transposed convs): 22: 128 x 128   23: 64 x 128 as two 32-row waves x 2   (lab: 21: 256 x 128, 8 waves)
        case 22: launch_bf3<2, 2, 2, 2, 1, 1, true, true>(a, nphase, st, a.rowph ? 0 : 1); break;
        case 23: launch_bf3<1, 2, 2, 2, 1, 1, true, true>(a, nphase, st, a.rowph ? 0 : 1); break;
        case 0: launch_bf3<2, 2, 2, 2, 1, 1, true>(a, nphase, st); break;
        case 3: launch_bf3<2, 2, 1, 2, 1, 1, true>(a, nphase, st); break;
#ifdef STS_EXPERIMENTS      // tiles no automatic choice selects (measured ties / losses, profiles/r03_*tile_sweep.log): lab build only
        case 24: if (a.Cin_pad % 32 == 0) { launch_bf3<2, 2, 4, 1, 2, 2, true>(a, nphase, st); break; } launch_bf3<2, 2, 2, 2, 1, 1, true>(a, nphase, st); break;
        case 21: launch_bf3<2, 2, 4, 2, 1>(a, nphase, st, 1); break;
        case 1: launch_bf3<2, 2, 1, 4, 1>(a, nphase, st); break;
        case 2: launch_bf3<2, 2, 2, 4, 1>(a, nphase, st); break;
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
        case 13: launch_bf3<1, 2, 1, 2, 2>(a, nphase, st); break;
#endif
        default: launch_bf3<1, 2, 1, 4, 1, 1, true>(a, nphase, st); break;       // 4: 32 x 256
    }
}

}  // namespace sts
