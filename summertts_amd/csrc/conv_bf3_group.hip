// conv_bf3_group.hip -- grouped launches of the split-operand convs: layer d of ALL ResBlock chains of a decoder stage in one grid
// (conv_bf3_group_kernel, conv_bf3_dev.hpp).  /root/reference/src/modules/ResBlock1.cpp:55-69, Generator_hifigan.cpp:151-175.
#include "conv_bf3_dev.hpp"

namespace sts {

#ifdef STS_TILE_TRACE
int tile_trace_bind_group(long long* buf, unsigned capacity_records) { return tile_trace_bind(buf, capacity_records); }
#endif


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
