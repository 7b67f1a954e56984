// Shared host/device helpers for the gfx950 latent-NeRF kernels.
// Compiled with -ffp-contract=off: every fused multiply-add in this library is written
// explicitly (fmaf / __builtin_fmaf) so that the discrete decisions of the ray march are
// reproducible against the CPU oracle (DESIGN.md "Arithmetic contract").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/lnerf_hip.h"

#define LNERF_WAVE 64

namespace lnerf {

void set_error(const char *fmt, ...);

inline hipStream_t as_stream(lnerf_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

#define LNERF_REQUIRE(cond, ...)                \
    do {                                        \
        if (!(cond)) {                          \
            ::lnerf::set_error(__VA_ARGS__);    \
            return LNERF_ERR_INVALID_ARG;       \
        }                                       \
    } while (0)

#define LNERF_CHECK_LAUNCH(name)                                                         \
    do {                                                                                 \
        hipError_t e__ = hipGetLastError();                                              \
        if (e__ != hipSuccess) {                                                         \
            ::lnerf::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));   \
            return LNERF_ERR_HIP;                                                        \
        }                                                                                \
    } while (0)

static inline int64_t div_up(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ uint32_t expand_bits(uint32_t v) {
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}
__device__ __forceinline__ uint32_t compact_bits(uint32_t x) {
    x &= 0x49249249u;
    x = (x | (x >> 2)) & 0xC30C30C3u;
    x = (x | (x >> 4)) & 0x0F00F00Fu;
    x = (x | (x >> 8)) & 0xFF0000FFu;
    x = (x | (x >> 16)) & 0x0000FFFFu;
    return x;
}
__device__ __forceinline__ uint32_t morton3d(uint32_t x, uint32_t y, uint32_t z) {
    return expand_bits(x) | (expand_bits(y) << 1) | (expand_bits(z) << 2);
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

// number of set bits of `mask` strictly below this lane (wave64)
__device__ __forceinline__ int mbcnt(unsigned long long mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

// ---- cross-lane data movement through DPP (data-parallel primitives): no LDS crossbar traffic.
// ctrl: 0x110+n row_shr:n (inside a 16-lane row), 0x138 wave_shr:1, 0x142 row_bcast:15, 0x143 row_bcast:31
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int old, int v) {
    return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xf, true);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true));
}
// value of lane-1 (lane 0 gets `fill`)
__device__ __forceinline__ int lane_prev_i(int v, int fill) {
    return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);
}

// inclusive wave scan (sum) over 64 lanes: 4 row_shr steps inside each 16-lane row, then the row
// totals are carried across rows with row_bcast:15 (rows 1,3) and row_bcast:31 (rows 2,3)
__device__ __forceinline__ float wave_inclusive_sum(float v) {
    v += dpp_f<0x111, 0xf>(v);
    v += dpp_f<0x112, 0xf>(v);
    v += dpp_f<0x114, 0xf>(v);
    v += dpp_f<0x118, 0xf>(v);
    v += dpp_f<0x142, 0xa>(v);
    v += dpp_f<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ int wave_inclusive_sum_i(int v) {
    v += dpp_i<0x111, 0xf>(0, v);
    v += dpp_i<0x112, 0xf>(0, v);
    v += dpp_i<0x114, 0xf>(0, v);
    v += dpp_i<0x118, 0xf>(0, v);
    v += dpp_i<0x142, 0xa>(0, v);
    v += dpp_i<0x143, 0xc>(0, v);
    return v;
}
// max over the wave of NON-NEGATIVE values (out-of-range DPP reads give 0), returned in every lane
__device__ __forceinline__ float wave_max_nonneg(float v) {
    v = fmaxf(v, dpp_f<0x111, 0xf>(v));
    v = fmaxf(v, dpp_f<0x112, 0xf>(v));
    v = fmaxf(v, dpp_f<0x114, 0xf>(v));
    v = fmaxf(v, dpp_f<0x118, 0xf>(v));
    v = fmaxf(v, dpp_f<0x142, 0xa>(v));
    v = fmaxf(v, dpp_f<0x143, 0xc>(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// sum over the wave, returned in every lane
__device__ __forceinline__ float wave_sum(float v) {
    v = wave_inclusive_sum(v);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float(((uint32_t)h) << 16); }
// round-to-nearest-even f32 -> bf16 (NaN preserved as quiet NaN)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

struct GridMeta {
    int num_levels;
    int offsets[LNERF_MAX_LEVELS + 1];
    float scales[LNERF_MAX_LEVELS];
    int res[LNERF_MAX_LEVELS];
    int blocked;   // layout of the levels larger than their table (grid.hip corner_rows): 0 = Instant-NGP vertex hash,
                   // 1 = LNERF_GRID_BLOCKED (4 x 2 x 2 vertex blocks hashed together), 2 = LNERF_GRID_TILED (dense index wrapped)
};

}  // namespace lnerf
