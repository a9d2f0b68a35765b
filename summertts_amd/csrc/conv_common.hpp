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

}  // namespace sts
