// conv_common.hpp -- device helpers shared by the matrix-core conv kernels (conv.hip, wn_layer.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include "kernels.hpp"
#include "devmath.hpp"

namespace sts {

constexpr int CK = 16;            // input channels staged per chunk
constexpr int MAX_HALO = 64;

typedef float f32x16 __attribute__((ext_vector_type(16)));

// compile-time loop: keeps every accumulator index static (a runtime-indexed ext_vector array
// would be demoted to scratch memory)
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Raw buffer descriptor (SRSRC): the hardware range-checks every access against num_records and
// returns 0 for anything outside -- zero padding without a single branch or select in the loop.
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float buf_load(rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}
constexpr unsigned kOOB = 0x7FFFFFF0u;   // byte offset that is out of range for every descriptor

// Wave-uniform values the compiler cannot always prove uniform (a segment table entry fetched with a vector load, a field of a
// group member selected through a pointer): forcing them into scalar registers keeps buffer descriptors and scalar offsets out of
// "waterfall" loops (a v_readfirstlane / compare / exec-mask loop around EVERY buffer load; round 3 found 65 of them in the
// grouped split-bf16 kernels).
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long uni(long v) {
    return (long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)v >> 32)) << 32) |
                  (unsigned)__builtin_amdgcn_readfirstlane((int)v));
}
template <typename T> __device__ __forceinline__ T* uni(T* p) { return (T*)uni((long)p); }

__device__ __forceinline__ int seg_start(const SegView& s, int b) { return (s.off ? s.off[b] : s.ioff) * s.scale + b * s.extra; }
__device__ __forceinline__ int seg_len(const SegView& s, int b) { return (s.off ? s.len[b] : s.ilen) * s.scale + s.extra; }

// Workgroup -> tile mapping, XCD-aware.  The grid is 1-D; hardware deals workgroup ids round-robin over the
// 8 XCDs (id & 7), each with a private L2.  All `ny` row tiles (and transposed-conv phases) that read the
// SAME input window are given to one XCD in consecutive dispatch slots, so the window is fetched from
// HBM once and re-read from that XCD's L2; units (column tile x utterance x group member) are dealt
// round-robin over the XCDs in launch order, which keeps the XCDs balanced when the members of a
// grouped launch differ in K.
struct TileId { int bx, by, bz; bool valid; };
__device__ __forceinline__ TileId map_tile(int nx, int ny, int nz) {
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int ul = slot / ny;
    TileId t;
    t.by = slot - ul * ny;
    const int unit = ul * 8 + xcd;
    t.valid = unit < nx * nz;
    t.bz = unit / nx;
    t.bx = unit - t.bz * nx;
    return t;
}
inline unsigned mapped_grid(int nx, int ny, int nz) { return (unsigned)(((long)nx * nz + 7) / 8 * 8 * ny); }

// scalar epilogue shared by both kernels (everything except the gate pairing)
__device__ __forceinline__ void epi_scalar(const ConvArgs& a, int row, size_t opos, float v) {
    switch (a.epi) {
        case EPI_STORE: a.y[(size_t)row * a.y_ld + opos] = v; break;
        case EPI_RESADD: a.y[(size_t)row * a.y_ld + opos] = v + a.res[(size_t)row * a.res_ld + opos]; break;
        case EPI_SUB: { float* p = a.y + (size_t)row * a.y_ld + opos; *p = *p - v; break; }
        case EPI_RESSKIP: {
            if (a.Cout != a.H && row < a.H) {
                float* p = a.y + (size_t)row * a.y_ld + opos; *p = *p + v;
            } else {
                int r = a.Cout != a.H ? row - a.H : row;
                float* p = a.aux + (size_t)r * a.aux_ld + opos;
                *p = (a.epi_flag & 1) ? v : *p + v;
            }
            break;
        }
        case EPI_TANH_PCM: {
            float t = tanh_ref(v);
            if (a.aux) a.aux[opos] = t;
            a.pcm[opos] = pcm_cast(t);
            break;
        }
        default: break;
    }
}

// 4 x 4 transpose inside every quad of lanes: before, lane 4m + c holds v[j] = M[row j][column 4m + c]; after, it holds
// v[j] = M[row c][column 4m + j] -- four CONSECUTIVE columns of one row.  Two exchange stages (lane bit 1, then lane bit 0) of
// DPP quad permutes + selects: ~16 VALU per 4 registers.  All 64 lanes must be active.
__device__ __forceinline__ float dpp_quad_2301(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true)); }
__device__ __forceinline__ float dpp_quad_1032(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true)); }
__device__ __forceinline__ void quad_transpose(float (&v)[4], const int lane) {
    const bool hi = (lane & 2) != 0, lo = (lane & 1) != 0;
    const float r0 = dpp_quad_2301(hi ? v[0] : v[2]), r1 = dpp_quad_2301(hi ? v[1] : v[3]);
    if (hi) { v[0] = r0; v[1] = r1; } else { v[2] = r0; v[3] = r1; }
    const float s0 = dpp_quad_1032(lo ? v[0] : v[1]), s1 = dpp_quad_1032(lo ? v[2] : v[3]);
    if (lo) { v[0] = s0; v[2] = s1; } else { v[1] = s0; v[3] = s1; }
}
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));      // 16-byte access with dword alignment (packed rows start anywhere)

// Epilogue on the accumulator registers (C/D layout of every 32x32 MFMA: col (time) = lane & 31,
// row (channel) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)).  The plain and the residual form -- every decoder trunk conv --
// take a branch-free path: a row tile's 16 bias values and a column tile's 16 residual values are requested together
// and waited for once.  (Measured: with the per-element epilogue switch each of a lane's 64 outputs paid its own dependent
// bias load -- ~31 us of a 57 us tile for a 3-tap 128-channel conv.)
// NTL: the residual is read with non-temporal loads (for data written by another CU of this XCD inside the same launch: a plain load could
// hit a stale line of this CU's vector L1).  Always false since the persistent stage kernel of round 3 was deleted (round 5)
// The residual operand of the plain / residual epilogue below (EPI_RESADD, unit output stride), requested AHEAD of the K loop: the tile's
// 16-byte residual loads then land under the MFMAs instead of costing the epilogue a memory round trip (round 4, tile trace: epilogue of a
// 128 x 128 tile 7.5 us without a residual, 11.5 us with one).  Same addresses, same predicates as the epilogue's own loads.
template <int MW, int NW, bool NTL = false>
__device__ __forceinline__ void tile_res_prefetch(const ConvArgs& a, f32x4u (&rv)[MW][NW][4], const int mbase, const int ncol0, const int l31,
                                                 const int half, const int n_count, const int out_len, const size_t out_base, const int phase) {
    const int out_off = a.out_off + phase;
    const int lane4 = l31 & 3, m4 = (l31 >> 2) * 4;
#pragma unroll
    for (int i = 0; i < MW; i++) {
        const bool rows_ok = mbase + i * 32 < a.Cout_pad;
#pragma unroll
        for (int q = 0; q < NW; q++) {
            const int n = ncol0 + q * 32 + m4;
            const int pos = n + out_off;
            const int klo = a.keep_hi > 0 ? a.keep_lo : 0, khi = a.keep_hi > 0 ? a.keep_hi : 0x7fffffff;
            const bool full = n + 3 < n_count && pos >= 0 && pos + 3 < out_len && n >= klo && n + 3 < khi;
            auto ok = [&](int e) { return n + e < n_count && pos + e >= 0 && pos + e < out_len && n + e >= klo && n + e < khi; };
            const bool any = n < n_count && pos + 3 >= 0 && pos < out_len && n + 3 >= klo && n < khi;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int row = mbase + i * 32 + 8 * g + 4 * half + lane4;
                rv[i][q][g] = f32x4u{0.f, 0.f, 0.f, 0.f};
                if (rows_ok && row < a.Cout && any) {
                    const float* rp = a.res + (size_t)row * a.res_ld + out_base + pos;
                    if (full) rv[i][q][g] = NTL ? __builtin_nontemporal_load((const f32x4u*)rp) : *(const f32x4u*)rp;
                    else { for (int e = 0; e < 4; e++) if (ok(e)) rv[i][q][g][e] = NTL ? __builtin_nontemporal_load(rp + e) : rp[e]; }
                }
            }
        }
    }
}

template <int MW, int NW, bool NTL, bool PRE, int PM, int PN>
__device__ __forceinline__ void tile_epilogue_impl(const ConvArgs& a, f32x16 (&acc)[MW][NW], const int mbase, const int ncol0, const int l31,
                                                  const int half, const int n_count, const int out_len, const size_t out_base, const int phase, const int b,
                                                  const f32x4u (&pre)[PM][PN][4], const bool use_pre) {
    const int out_off = a.out_off + phase;
    if ((a.epi == EPI_STORE || a.epi == EPI_RESADD) && a.out_stride == 1) {
        // Plain convs (every ResBlock conv of the trunk).  Round 3 (tools/tile_trace.py): with one dword store and one dword residual
        // load per accumulator register the epilogue of a 128 x 128 tile took 16-24 us -- store-ISSUE-bound, 25-45 % of a tile's
        // life.  A lane's 4 registers of a row group are 4 consecutive ROWS of one column; after a 4 x 4 transpose inside the lane
        // quads they are 4 consecutive COLUMNS of one row, so a tile leaves through 16-byte stores (and its residual arrives through
        // 16-byte loads): a quarter of the memory instructions, each covering four 128-byte row segments.  Same arithmetic per value.
        const int lane4 = l31 & 3, m4 = (l31 >> 2) * 4;
        static_for<0, MW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const bool rows_ok = mbase + i * 32 < a.Cout_pad;
            float bv[4], uv[4];
            int rowv[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                rowv[g] = mbase + i * 32 + 8 * g + 4 * half + lane4;
                bv[g] = (rows_ok && a.bias) ? a.bias[rowv[g]] : 0.f;             // bias is padded to Cout_pad
                uv[g] = (rows_ok && a.ubias && rowv[g] < a.Cout) ? a.ubias[(size_t)rowv[g] * a.ubias_ld + b] : 0.f;
            }
            static_for<0, NW>([&](auto qc) {
                constexpr int q = decltype(qc)::value;
                const int n = ncol0 + q * 32 + m4;
                const int pos = n + out_off;            // output position of column n (out_off != 0: a window that keeps only part of its columns)
                const int klo = a.keep_hi > 0 ? a.keep_lo : 0, khi = a.keep_hi > 0 ? a.keep_hi : 0x7fffffff;
                const bool full = n + 3 < n_count && pos >= 0 && pos + 3 < out_len && n >= klo && n + 3 < khi;
                auto ok = [&](int e) { return n + e < n_count && pos + e >= 0 && pos + e < out_len && n + e >= klo && n + e < khi; };
                const bool any = n < n_count && pos + 3 >= 0 && pos < out_len && n + 3 >= klo && n < khi;
                f32x4u rv[4];
                bool have = false;
                if constexpr (PRE) {
                    if (a.epi == EPI_RESADD && use_pre) {
                        have = true;
#pragma unroll
                        for (int g = 0; g < 4; g++) rv[g] = pre[i][q][g];
                    }
                }
                if (a.epi == EPI_RESADD && !have) {
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        rv[g] = f32x4u{0.f, 0.f, 0.f, 0.f};
                        if (rows_ok && rowv[g] < a.Cout && any) {
                            const float* rp = a.res + (size_t)rowv[g] * a.res_ld + out_base + pos;
                            if (full) rv[g] = NTL ? __builtin_nontemporal_load((const f32x4u*)rp) : *(const f32x4u*)rp;
                            else { for (int e = 0; e < 4; e++) if (ok(e)) rv[g][e] = NTL ? __builtin_nontemporal_load(rp + e) : rp[e]; }
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    float w[4] = {acc[i][q][4 * g], acc[i][q][4 * g + 1], acc[i][q][4 * g + 2], acc[i][q][4 * g + 3]};
                    quad_transpose(w, l31);
                    if (rows_ok && rowv[g] < a.Cout && any) {
                        f32x4u o;
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float r = a.epi == EPI_RESADD ? rv[g][e] : 0.f;
                            o[e] = a.ubias ? ((w[e] + bv[g]) + uv[g]) + r : (w[e] + bv[g]) + r;
                        }
                        float* yp = a.y + (size_t)rowv[g] * a.y_ld + out_base + pos;
                        if (full) *(f32x4u*)yp = o;
                        else { for (int e = 0; e < 4; e++) if (ok(e)) yp[e] = o[e]; }
                    }
                }
            });
        });
        return;
    }
    if (a.rowph && a.epi == EPI_STORE) {
        // Polyphase transposed conv whose packed rows interleave the phases (ConvArgs::rowph): tile row rho = cout * s + phase.  A lane's four
        // consecutive accumulator rows (4 half + 0..3 of every 8) are four consecutive OUTPUT POSITIONS of one channel for s = 4 / 8 (two
        // positions of two channels for s = 2), columns n -> positions n s + phase: a wave's store covers contiguous runs of each row
        // (whole 32-byte sectors) instead of one dword per lane at a stride of s dwords.  Same value per element as the form below.
        const int s = a.out_stride, sh = s == 8 ? 3 : (s == 4 ? 2 : 1);
        static_for<0, MW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (mbase + i * 32 < a.Cout_pad) {
                f32x4u bv[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int rho0 = mbase + i * 32 + 8 * g + 4 * half;
                    bv[g] = a.bias ? *(const f32x4u*)(a.bias + rho0) : f32x4u{0.f, 0.f, 0.f, 0.f};        // bias per merged row, padded to Cout_pad
                }
                static_for<0, NW>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    const int n = ncol0 + q * 32 + l31;
                    if (n < n_count) {
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            const int rho0 = mbase + i * 32 + 8 * g + 4 * half;
                            if (rho0 >= a.Cout) continue;
                            const f32x4u v = {acc[i][q][4 * g] + bv[g][0], acc[i][q][4 * g + 1] + bv[g][1], acc[i][q][4 * g + 2] + bv[g][2], acc[i][q][4 * g + 3] + bv[g][3]};
                            const int co = rho0 >> sh;
                            const int pos = n * s + a.out_off + (rho0 & (s - 1));
                            if (s >= 4) {
                                float* yp = a.y + (size_t)co * a.y_ld + out_base + pos;
                                if (pos >= 0 && pos + 3 < out_len) *(f32x4u*)yp = v;
                                else { for (int e = 0; e < 4; e++) if (pos + e >= 0 && pos + e < out_len) yp[e] = v[e]; }
                            } else {
#pragma unroll
                                for (int c2 = 0; c2 < 2; c2++) {
                                    float* yp = a.y + (size_t)(co + c2) * a.y_ld + out_base + pos;
                                    if (pos >= 0 && pos + 1 < out_len) *(f32x2u*)yp = f32x2u{v[2 * c2], v[2 * c2 + 1]};
                                    else { for (int e = 0; e < 2; e++) if (pos + e >= 0 && pos + e < out_len) yp[e] = v[2 * c2 + e]; }
                                }
                            }
                        }
                    }
                });
            }
        });
        return;
    }
    if (a.epi == EPI_STORE || a.epi == EPI_RESADD) {
        static_for<0, MW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (mbase + i * 32 < a.Cout_pad) {
                float bv[16], uv[16];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int rowp = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    bv[r] = a.bias ? a.bias[rowp] : 0.f;            // bias is padded to Cout_pad
                    uv[r] = (a.ubias && rowp < a.Cout) ? a.ubias[(size_t)rowp * a.ubias_ld + b] : 0.f;
                }
                static_for<0, NW>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    const int n = ncol0 + q * 32 + l31;
                    const int pos = n * a.out_stride + out_off;
                    if (n < n_count && pos >= 0 && pos < out_len) {
                        const size_t opos = out_base + (size_t)pos;
                        float rv[16];
                        if (a.epi == EPI_RESADD) {
#pragma unroll
                            for (int r = 0; r < 16; r++) {
                                const int rowp = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                                rv[r] = rowp < a.Cout ? a.res[(size_t)rowp * a.res_ld + opos] : 0.f;
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 16; r++) rv[r] = 0.f;
                        }
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int rowp = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (rowp < a.Cout) a.y[(size_t)rowp * a.y_ld + opos] = a.ubias ? ((acc[i][q][r] + bv[r]) + uv[r]) + rv[r] : (acc[i][q][r] + bv[r]) + rv[r];
                        }
                    }
                });
            }
        });
        return;
    }
    if (a.epi == EPI_GATE && a.out_stride == 1) {
        // Round 6: the WaveNet gate with 16-byte stores.  Registers r and r + 8 of a lane are tile rows (c, c + 16) = the tanh and the sigmoid
        // pre-activation of gated channel 16 (tile) + c; the lane's 8 gated values are channels {4 half + 0..3} and {8 + 4 half + 0..3} of ONE
        // column -- two groups of 4 consecutive channels, which the in-quad transpose turns into 4 consecutive columns of one channel each
        // (conv_common.hpp quad_transpose, as the plain epilogue above).  Same arithmetic per value as the scalar form below.
        const int lane4 = l31 & 3, m4 = (l31 >> 2) * 4;
        static_for<0, MW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (mbase + i * 32 < a.Cout_pad) {
                float bt[8], bs[8], ut[8], us[8];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int rowp = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    bt[r] = a.bias ? a.bias[rowp] : 0.f;
                    bs[r] = a.bias ? a.bias[rowp + 16] : 0.f;
                    ut[r] = a.ubias ? a.ubias[(size_t)rowp * a.ubias_ld + b] : 0.f;
                    us[r] = a.ubias ? a.ubias[(size_t)(rowp + 16) * a.ubias_ld + b] : 0.f;
                }
                static_for<0, NW>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    float gv[8];
#pragma unroll
                    for (int r = 0; r < 8; r++)
                        gv[r] = tanh_ref(a.ubias ? (acc[i][q][r] + bt[r]) + ut[r] : acc[i][q][r] + bt[r]) *
                                sigmoid_ref(a.ubias ? (acc[i][q][r + 8] + bs[r]) + us[r] : acc[i][q][r + 8] + bs[r]);
                    const int n = ncol0 + q * 32 + m4;
                    const int pos = n + out_off;
                    const bool full = n + 3 < n_count && pos >= 0 && pos + 3 < out_len;
                    const bool any = n < n_count && pos + 3 >= 0 && pos < out_len;
#pragma unroll
                    for (int g = 0; g < 2; g++) {
                        float w[4] = {gv[4 * g], gv[4 * g + 1], gv[4 * g + 2], gv[4 * g + 3]};
                        quad_transpose(w, l31);
                        const int rowp = mbase + i * 32 + 8 * g + 4 * half + lane4;        // tile row of the tanh half
                        const int ch = (rowp >> 5) * 16 + (rowp & 15);
                        if (ch < a.H && any) {
                            float* yp = a.y + (size_t)ch * a.y_ld + out_base + pos;
                            if (full) *(f32x4u*)yp = f32x4u{w[0], w[1], w[2], w[3]};
                            else { for (int e = 0; e < 4; e++) if (n + e < n_count && pos + e >= 0 && pos + e < out_len) yp[e] = w[e]; }
                        }
                    }
                });
            }
        });
        return;
    }
    if (a.epi == EPI_GATE) {
        // rows r and r + 8 of a lane's 16 accumulator rows are tile rows (c, c + 16): the tanh and the sigmoid pre-activation of one
        // channel (model.hip gate_perm_row); a row tile's 16 bias values are requested together
        static_for<0, MW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (mbase + i * 32 < a.Cout_pad) {
                float bt[8], bs[8], ut[8], us[8];
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int rowp = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    bt[r] = a.bias ? a.bias[rowp] : 0.f;
                    bs[r] = a.bias ? a.bias[rowp + 16] : 0.f;
                    ut[r] = a.ubias ? a.ubias[(size_t)rowp * a.ubias_ld + b] : 0.f;
                    us[r] = a.ubias ? a.ubias[(size_t)(rowp + 16) * a.ubias_ld + b] : 0.f;
                }
                static_for<0, NW>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    const int n = ncol0 + q * 32 + l31;
                    const int pos = n * a.out_stride + out_off;
                    if (n < n_count && pos >= 0 && pos < out_len) {
                        const size_t opos = out_base + (size_t)pos;
#pragma unroll
                        for (int r = 0; r < 8; r++) {
                            const int rowp = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            const int ch = (rowp >> 5) * 16 + (rowp & 15);
                            if (ch < a.H) a.y[(size_t)ch * a.y_ld + opos] = tanh_ref(a.ubias ? (acc[i][q][r] + bt[r]) + ut[r] : acc[i][q][r] + bt[r]) *
                                                                            sigmoid_ref(a.ubias ? (acc[i][q][r + 8] + bs[r]) + us[r] : acc[i][q][r + 8] + bs[r]);
                        }
                    }
                });
            }
        });
        return;
    }
    if ((a.epi == EPI_RESSKIP || a.epi == EPI_SUB) && a.out_stride == 1) {
        // Round 6: the flow's read-modify-write epilogues (WN res / skip split, coupling "x1 -= m") through 16-byte accesses: after the in-quad
        // transpose a lane holds 4 consecutive columns of one row, so the old values arrive in one 16-byte load per 4 registers and leave in one
        // store (the scalar form below: 16 dword loads + 16 dword stores per accumulator tile and lane -- these launches were epilogue-bound:
        // the res/skip conv has 12 K steps).  Which tensor a row belongs to (h / skip, x1) is a property of the row: uniform per store.
        const int lane4 = l31 & 3, m4 = (l31 >> 2) * 4;
        static_for<0, MW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (mbase + i * 32 < a.Cout_pad) {
                float bv[4], uv[4];
                float* rowptr[4]; bool ld_old[4], rok[4];
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int rowp = mbase + i * 32 + 8 * g + 4 * half + lane4;
                    bv[g] = a.bias ? a.bias[rowp] : 0.f;
                    uv[g] = (a.ubias && rowp < a.Cout) ? a.ubias[(size_t)rowp * a.ubias_ld + b] : 0.f;
                    rok[g] = rowp < a.Cout;
                    ld_old[g] = true;
                    if (a.epi == EPI_SUB || (a.Cout != a.H && rowp < a.H)) rowptr[g] = a.y + (size_t)rowp * a.y_ld;
                    else { rowptr[g] = a.aux + (size_t)(a.Cout != a.H ? rowp - a.H : rowp) * a.aux_ld; ld_old[g] = !(a.epi_flag & 1); }
                }
                static_for<0, NW>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    const int n = ncol0 + q * 32 + m4;
                    const int pos = n + out_off;
                    const bool full = n + 3 < n_count && pos >= 0 && pos + 3 < out_len;
                    const bool any = n < n_count && pos + 3 >= 0 && pos < out_len;
                    auto ok = [&](int e) { return n + e < n_count && pos + e >= 0 && pos + e < out_len; };
                    f32x4u old[4];
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        old[g] = f32x4u{0.f, 0.f, 0.f, 0.f};
                        if (rok[g] && ld_old[g] && any) {
                            const float* rp = rowptr[g] + out_base + pos;
                            if (full) old[g] = *(const f32x4u*)rp;
                            else { for (int e = 0; e < 4; e++) if (ok(e)) old[g][e] = rp[e]; }
                        }
                    }
#pragma unroll
                    for (int g = 0; g < 4; g++) {
                        float w[4] = {acc[i][q][4 * g], acc[i][q][4 * g + 1], acc[i][q][4 * g + 2], acc[i][q][4 * g + 3]};
                        quad_transpose(w, l31);
                        if (rok[g] && any) {
                            f32x4u o;
#pragma unroll
                            for (int e = 0; e < 4; e++) {
                                const float v = a.ubias ? (w[e] + bv[g]) + uv[g] : w[e] + bv[g];
                                o[e] = a.epi == EPI_SUB ? old[g][e] - v : old[g][e] + v;
                            }
                            float* wp = rowptr[g] + out_base + pos;
                            if (full) *(f32x4u*)wp = o;
                            else { for (int e = 0; e < 4; e++) if (ok(e)) wp[e] = o[e]; }
                        }
                    }
                });
            }
        });
        return;
    }
    if (a.epi == EPI_RESSKIP || a.epi == EPI_SUB) {
        // read-modify-write epilogues of the flow (WN res/skip split, coupling "x1 -= m"): a column tile's 16 old values are
        // requested together, then updated and stored
        static_for<0, MW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (mbase + i * 32 < a.Cout_pad) {
                float bv[16], uv[16];
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int rowp = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    bv[r] = a.bias ? a.bias[rowp] : 0.f;
                    uv[r] = (a.ubias && rowp < a.Cout) ? a.ubias[(size_t)rowp * a.ubias_ld + b] : 0.f;
                }
                static_for<0, NW>([&](auto qc) {
                    constexpr int q = decltype(qc)::value;
                    const int n = ncol0 + q * 32 + l31;
                    const int pos = n * a.out_stride + out_off;
                    if (n < n_count && pos >= 0 && pos < out_len) {
                        const size_t opos = out_base + (size_t)pos;
                        float* p[16]; float old[16];
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int rowp = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            bool ld = true;
                            if (a.epi == EPI_SUB || (a.Cout != a.H && rowp < a.H)) p[r] = a.y + (size_t)rowp * a.y_ld + opos;
                            else { p[r] = a.aux + (size_t)(a.Cout != a.H ? rowp - a.H : rowp) * a.aux_ld + opos; ld = !(a.epi_flag & 1); }
                            old[r] = (rowp < a.Cout && ld) ? *p[r] : 0.f;
                        }
#pragma unroll
                        for (int r = 0; r < 16; r++) {
                            const int rowp = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            const float v = a.ubias ? (acc[i][q][r] + bv[r]) + uv[r] : acc[i][q][r] + bv[r];
                            if (rowp < a.Cout) *p[r] = a.epi == EPI_SUB ? old[r] - v : old[r] + v;
                        }
                    }
                });
            }
        });
        return;
    }
    static_for<0, MW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        static_for<0, NW>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const int n = ncol0 + q * 32 + l31;
            const int pos = n * a.out_stride + out_off;
            if (mbase + i * 32 < a.Cout_pad && n < n_count && pos >= 0 && pos < out_len) {
                const size_t opos = out_base + (size_t)pos;
                static_for<0, 16>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    const int rowp = mbase + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (rowp < a.Cout) {
                        float v = acc[i][q][r];
                        if (a.bias) v += a.bias[rowp];
                        if (a.ubias) v += a.ubias[(size_t)rowp * a.ubias_ld + b];
                        epi_scalar(a, rowp, opos, v);
                    }
                });
            }
        });
    });
}

template <int MW, int NW, bool NTL = false>
__device__ __forceinline__ void tile_epilogue(const ConvArgs& a, f32x16 (&acc)[MW][NW], const int mbase, const int ncol0, const int l31,
                                             const int half, const int n_count, const int out_len, const size_t out_base, const int phase, const int b) {
    const f32x4u none[1][1][4] = {};
    tile_epilogue_impl<MW, NW, NTL, false, 1, 1>(a, acc, mbase, ncol0, l31, half, n_count, out_len, out_base, phase, b, none, false);
}
// ... with the residual tile already in registers (tile_res_prefetch) when use_pre
template <int MW, int NW, bool NTL = false>
__device__ __forceinline__ void tile_epilogue_pre(const ConvArgs& a, f32x16 (&acc)[MW][NW], const int mbase, const int ncol0, const int l31,
                                                 const int half, const int n_count, const int out_len, const size_t out_base, const int phase, const int b,
                                                 const f32x4u (&pre)[MW][NW][4], const bool use_pre) {
    tile_epilogue_impl<MW, NW, NTL, true, MW, NW>(a, acc, mbase, ncol0, l31, half, n_count, out_len, out_base, phase, b, pre, use_pre);
}

}  // namespace sts
