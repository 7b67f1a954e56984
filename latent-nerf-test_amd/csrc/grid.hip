// H5/H6 multiresolution hash grid (Instant-NGP encoding, F = 2 features per vertex).
//
// Forward is the roofline kernel of the path: per sample and level it gathers
// 8 vertices x 2 features.  One thread handles one (sample, level); a wavefront handles 64
// consecutive samples of ONE level, so its 8 gather instructions hit one level's table and its
// output is 512 contiguous bytes (level-major feature layout).
//
// variant 0: blockIdx.y = level.
// variant 1: XCD-aware.  Workgroups are dealt round-robin over the 8 XCDs (observed, used for
//            speed only -- correctness never depends on it), so workgroup b serves levels
//            {b % 8, b % 8 + 8, ...}: each XCD's private 4 MiB L2 then only ever holds the
//            tables of its own levels instead of all 16.
#include "common.h"
#include "adam_shared.h"
#include "mlp_shared.h"

#include <stdlib.h>
#include <string.h>

namespace lnerf {

__device__ __forceinline__ uint32_t grid_index(uint32_t x, uint32_t y, uint32_t z, uint32_t res, uint32_t hsize) {
    // dense while the (res+1)^3 vertex lattice fits the level, spatial hash otherwise.
    // res/hsize are wave-uniform, so both branches below are scalar branches.
    const uint32_t stride = res + 1;
    const uint64_t cube = (uint64_t)stride * stride * stride;  // (res+1) <= 2^20: no overflow
    if (cube <= (uint64_t)hsize) return x + y * stride + z * stride * stride;  // < hsize already
    const uint32_t index = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);
    if ((hsize & (hsize - 1u)) == 0u) return index & (hsize - 1u);
    return index % hsize;
}

// rows of the 8 vertices of a cell at once: the two integer multiplies of the spatial hash (quarter-rate
// VALU) are shared by all corners ((y+1)*P == y*P + P mod 2^32), dense levels add strides to one base.
// blocked (LNERF_GRID_BLOCKED, an opt-in layout of the HASHED levels, not Instant-NGP's): the lattice is cut into blocks
// of 4 x 2 x 2 vertices, the BLOCK coordinate is hashed and a block's 16 rows are consecutive --
//     row = (hash(x >> 2, y >> 1, z >> 1) mod (hsize / 16)) * 16 + (x & 3) + 4 (y & 1) + 8 (z & 1)
// -- so that a block is one 64-byte line of the bf16 table: a cell's 8 vertices touch 1.25 x 1.5 x 1.5 = 2.8 lines on
// average instead of 4.25 (x pairs share a line either way; here y and z neighbours do half the time).
// tiled (LNERF_GRID_TILED: `gridtype = "tiled"` of the upstream encoder): a level too large for its
// table wraps its DENSE index instead of hashing the vertex --
//     row = (x + y (res + 1) + z (res + 1)^2  mod 2^32)  mod hsize
// -- x-neighbours stay neighbours, whole y / z slabs alias each other.
__device__ __forceinline__ void corner_rows(uint32_t gx, uint32_t gy, uint32_t gz, uint32_t res, uint32_t hsize,
                                            uint32_t row[8], int layout = 0) {
    const bool blocked = layout == 1;
    const uint32_t stride = res + 1;
    const uint64_t cube = (uint64_t)stride * stride * stride;
    if (cube <= (uint64_t)hsize) {  // wave-uniform
        // (a dense level has stride^3 <= hsize < 2^31: all factors below 2^24 -- full-rate 24-bit multiplies)
        const uint32_t s2 = stride * stride;
        const uint32_t base = gx + __umul24(gy, stride) + __umul24(gz, s2);
#pragma unroll
        for (int c = 0; c < 8; ++c) row[c] = base + (c & 1) + ((c >> 1) & 1) * stride + ((c >> 2) & 1) * s2;
        return;
    }
    if (layout == 2) {  // wave-uniform: tiled
        const uint32_t s2 = stride * stride;   // (uint32 wrap-around, as the upstream's index arithmetic)
        const uint32_t base = gx + gy * stride + gz * s2;
        if ((hsize & (hsize - 1u)) == 0u) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                row[c] = (base + (c & 1) + ((c >> 1) & 1) * stride + ((c >> 2) & 1) * s2) & (hsize - 1u);
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                row[c] = (base + (c & 1) + ((c >> 1) & 1) * stride + ((c >> 2) & 1) * s2) % hsize;
        }
        return;
    }
    if (blocked) {  // wave-uniform
        const uint32_t nblk = hsize >> 4;
        const uint32_t x1 = gx + 1u;
        const uint32_t hx[2] = {gx >> 2, x1 >> 2};
        const uint32_t y0 = (gy >> 1) * 2654435761u, z0 = (gz >> 1) * 805459861u;
        // (y + 1) >> 1 is the next block exactly when y is odd
        const uint32_t hy[2] = {y0, (gy & 1u) ? y0 + 2654435761u : y0};
        const uint32_t hz[2] = {z0, (gz & 1u) ? z0 + 805459861u : z0};
        const uint32_t wx[2] = {gx & 3u, x1 & 3u};
        const uint32_t wy[2] = {(gy & 1u) << 2, ((gy + 1u) & 1u) << 2};
        const uint32_t wz[2] = {(gz & 1u) << 3, ((gz + 1u) & 1u) << 3};
        if ((nblk & (nblk - 1u)) == 0u) {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                row[c] = (((hx[c & 1] ^ hy[(c >> 1) & 1] ^ hz[(c >> 2) & 1]) & (nblk - 1u)) << 4) |
                         (wx[c & 1] | wy[(c >> 1) & 1] | wz[(c >> 2) & 1]);
        } else {
#pragma unroll
            for (int c = 0; c < 8; ++c)
                row[c] = (((hx[c & 1] ^ hy[(c >> 1) & 1] ^ hz[(c >> 2) & 1]) % nblk) << 4) |
                         (wx[c & 1] | wy[(c >> 1) & 1] | wz[(c >> 2) & 1]);
        }
        return;
    }
    const uint32_t hx[2] = {gx, gx + 1u};
    const uint32_t y0 = gy * 2654435761u, z0 = gz * 805459861u;
    const uint32_t hy[2] = {y0, y0 + 2654435761u};
    const uint32_t hz[2] = {z0, z0 + 805459861u};
    if ((hsize & (hsize - 1u)) == 0u) {
        const uint32_t mask = hsize - 1u;
#pragma unroll
        for (int c = 0; c < 8; ++c) row[c] = (hx[c & 1] ^ hy[(c >> 1) & 1] ^ hz[(c >> 2) & 1]) & mask;
    } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) row[c] = (hx[c & 1] ^ hy[(c >> 1) & 1] ^ hz[(c >> 2) & 1]) % hsize;
    }
}

template <typename T> struct Feat2;
template <> struct Feat2<float> {
    static __device__ __forceinline__ float2 load(const float *base, uint32_t row) {
        return reinterpret_cast<const float2 *>(base)[row];
    }
    static __device__ __forceinline__ void store(float *base, int64_t i, float a, float b) {
        reinterpret_cast<float2 *>(base)[i] = make_float2(a, b);
    }
};
template <> struct Feat2<uint16_t> {  // bf16 pairs in one dword
    static __device__ __forceinline__ float2 load(const uint16_t *base, uint32_t row) {
        const uint32_t v = reinterpret_cast<const uint32_t *>(base)[row];
        return make_float2(__uint_as_float(v << 16), __uint_as_float(v & 0xFFFF0000u));
    }
    static __device__ __forceinline__ void store(uint16_t *base, int64_t i, float a, float b) {
        reinterpret_cast<uint32_t *>(base)[i] = (uint32_t)f32_to_bf16(a) | ((uint32_t)f32_to_bf16(b) << 16);
    }
};

struct LevelPos {
    uint32_t gx, gy, gz;
    float fx, fy, fz;
};

__device__ __forceinline__ LevelPos level_pos_xyz(float x, float y, float z, float bound, float scale) {
    // x01 = (x + bound) / (2 bound); pos = x01 * scale + 0.5   (op order = oracle grid_encode)
    const float two_b = 2.0f * bound;
    float px = x + bound, py = y + bound, pz = z + bound;
    if ((__float_as_uint(two_b) & 0x007FFFFFu) == 0u) {  // power of two (wave-uniform): x / 2^k == x * 2^-k exactly
        const float r = 1.0f / two_b;
        px = px * r; py = py * r; pz = pz * r;
    } else {
        px = px / two_b; py = py / two_b; pz = pz / two_b;
    }
    px = px * scale; py = py * scale; pz = pz * scale;
    px = px + 0.5f; py = py + 0.5f; pz = pz + 0.5f;
    const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
    LevelPos r;
    r.gx = (uint32_t)(int)flx; r.gy = (uint32_t)(int)fly; r.gz = (uint32_t)(int)flz;
    r.fx = px - flx; r.fy = py - fly; r.fz = pz - flz;
    return r;
}
__device__ __forceinline__ LevelPos level_pos(const float *__restrict__ xyzs, int64_t m, float bound, float scale) {
    return level_pos_xyz(xyzs[m * 3], xyzs[m * 3 + 1], xyzs[m * 3 + 2], bound, scale);
}

// maps a workgroup to (level, first tile, tile step)
struct TileMap {
    int level;
    int64_t tile0, tstep;
    bool ok;
};
__device__ __forceinline__ TileMap tile_map(int variant, int L) {
    TileMap t;
    if (variant == 0) {
        t.level = blockIdx.y;
        t.tile0 = blockIdx.x;
        t.tstep = gridDim.x;
        t.ok = true;
    } else {
        // 1-D grid, gridDim.x = 8 * per_xcd.  slot = position inside the XCD's share.
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3;
        const int lv_per_xcd = (L + 7) >> 3;  // levels served by one XCD
        const int li = slot % lv_per_xcd;
        t.level = xcd + 8 * li;
        t.tile0 = slot / lv_per_xcd;
        t.tstep = per_xcd / lv_per_xcd;
        t.ok = (t.level < L) && (t.tile0 < t.tstep);
    }
    return t;
}

// Runs of samples that sit in the same grid cell (lanes = consecutive samples of a ray: on coarse levels long
// runs share all 8 vertices).  `start` = first lane of this lane's run, `tail` = this lane is the last lane of its
// run.  Computed once per (wave, level) from the cell coordinates; every lane of the wave must call it.
struct RunInfo {
    int start;
    bool tail;
    unsigned long long heads;  // wave-uniform: bit i = lane i starts a run (bit 0 always set)
};
__device__ __forceinline__ RunInfo wave_cell_runs(uint32_t gx, uint32_t gy, uint32_t gz, bool valid, int lane) {
    const int px = lane_prev_i((int)gx, -1), py = lane_prev_i((int)gy, -1), pz = lane_prev_i((int)gz, -1);
    const int pv = lane_prev_i((int)valid, 0);
    const bool head = (lane == 0) || px != (int)gx || py != (int)gy || pz != (int)gz || !valid || !pv;
    const unsigned long long H = __ballot(head);  // bit 0 is always set
    RunInfo r;
    r.start = 63 - __clzll((long long)(H & (~0ull >> (63 - lane))));
    r.tail = (lane == 63) || ((H >> (lane + 1)) & 1ull);
    r.heads = H;
    return r;
}

// the vertex values of one cell as raw dwords: 8 (bf16 pairs) or 16 (f32 pairs).  load_pair fetches two consecutive
// table rows with one load (x-adjacent vertices are adjacent rows on dense levels, and on hashed levels when x is
// even: row(x+1) = row(x) ^ 1): half the cache accesses of those lookups
template <typename TT> struct CellRaw;
template <> struct CellRaw<uint16_t> {
    uint32_t d[8];
    __device__ __forceinline__ void load_pair(const uint16_t *lt, uint32_t row, int c) {  // rows row, row+1 -> c, c+1
        const uint2 v = *reinterpret_cast<const uint2 *>(lt + (int64_t)row * 2);
        d[c] = v.x; d[c + 1] = v.y;
    }
    __device__ __forceinline__ void load_one(const uint16_t *lt, uint32_t row, int c) {
        d[c] = reinterpret_cast<const uint32_t *>(lt)[row];
    }
    // rows r0, r1 of ONE aligned group of four rows (16 bytes) with one load -> c, c + 1
    __device__ __forceinline__ void load_quad(const uint16_t *lt, uint32_t r0, uint32_t r1, int c) {
        const uint4 v = *reinterpret_cast<const uint4 *>(lt + (int64_t)(r0 & ~3u) * 2);
        const uint32_t k0 = r0 & 3u, k1 = r1 & 3u;
        d[c] = (k0 & 2u) ? ((k0 & 1u) ? v.w : v.z) : ((k0 & 1u) ? v.y : v.x);
        d[c + 1] = (k1 & 2u) ? ((k1 & 1u) ? v.w : v.z) : ((k1 & 1u) ? v.y : v.x);
    }
    static constexpr bool kHasQuad = true;
    __device__ __forceinline__ void swap_pair(int c) { const uint32_t t = d[c]; d[c] = d[c + 1]; d[c + 1] = t; }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = 0u;
    }
    __device__ __forceinline__ void take_from_lane(int src) {  // every lane reads lane `src`'s cell
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)d[i]);
    }
    __device__ __forceinline__ float2 get(int c) const {
        return make_float2(__uint_as_float(d[c] << 16), __uint_as_float(d[c] & 0xFFFF0000u));
    }
};
template <> struct CellRaw<float> {
    float2 d[8];
    __device__ __forceinline__ void load_pair(const float *lt, uint32_t row, int c) {
        const float4 v = *reinterpret_cast<const float4 *>(lt + (int64_t)row * 2);  // dword-aligned 16-byte load
        d[c] = make_float2(v.x, v.y); d[c + 1] = make_float2(v.z, v.w);
    }
    __device__ __forceinline__ void load_one(const float *lt, uint32_t row, int c) {
        d[c] = reinterpret_cast<const float2 *>(lt)[row];
    }
    __device__ __forceinline__ void load_quad(const float *, uint32_t, uint32_t, int) {}   // (32 bytes: not used)
    static constexpr bool kHasQuad = false;
    __device__ __forceinline__ void swap_pair(int c) { const float2 t = d[c]; d[c] = d[c + 1]; d[c + 1] = t; }
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < 8; ++i) d[i] = make_float2(0.f, 0.f);
    }
    __device__ __forceinline__ void take_from_lane(int src) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            d[i].x = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(d[i].x)));
            d[i].y = __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(d[i].y)));
        }
    }
    __device__ __forceinline__ float2 get(int c) const { return d[c]; }
};

// variant 2: every XCD serves a fixed SET of levels (workgroups are dealt round-robin over the 8 XCDs -- observed, used
// for speed only): the 4 MiB L2 of an XCD then holds the whole table of its one fine level (2 MiB bf16) instead of a
// sixth of all sixteen, and the gather -- bound by the L1's miss concurrency x the latency of a miss -- waits for L2 hits
// instead of Infinity-Cache hits.  The sets are balanced on the host from a per-level cost estimate.
struct XcdPlan {
    int n[8];
    int lv[8][LNERF_MAX_LEVELS / 8 + 2];
};

template <typename TT, typename TO>
__global__ void __launch_bounds__(256)
k_grid_forward(const float *__restrict__ xyzs, float bound, const TT *__restrict__ table, GridMeta meta, int64_t m_host,
               const int32_t *__restrict__ m_dev, int64_t level_stride, TO *__restrict__ feat, int variant,
               int pair_loads, int dedup_max_res, XcdPlan plan) {
#ifndef LNERF_EXPERIMENTS
    variant = 0;   // (the XCD-pinned mappings 1 / 2 are compiled into experiment builds only)
#endif
    int64_t M = m_host;
    if (m_dev) { const int64_t md = *m_dev; M = md < M ? md : M; }
    TileMap tm = tile_map(variant == 2 ? 0 : variant, meta.num_levels);
    int n_lv = 1;
    const int xcd = blockIdx.x & 7;
    if (variant == 2) {
        n_lv = plan.n[xcd];
        tm.tile0 = blockIdx.x >> 3;
        tm.tstep = gridDim.x >> 3;
        tm.ok = true;
    }
    if (!tm.ok) return;
  for (int li = 0; li < n_lv; ++li) {
    const int l = variant == 2 ? plan.lv[xcd][li] : tm.level;
    const float scale = meta.scales[l];
    const uint32_t res = (uint32_t)meta.res[l];
    const uint32_t off = (uint32_t)meta.offsets[l];
    const uint32_t hsize = (uint32_t)(meta.offsets[l + 1] - meta.offsets[l]);
    const TT *lt = table + (int64_t)off * 2;
    const bool dense = (uint64_t)(res + 1) * (res + 1) * (res + 1) <= (uint64_t)hsize;  // wave-uniform
    const bool pow2 = (hsize & (hsize - 1u)) == 0u;
    // Coarse levels: the 64 lanes of a wave are consecutive samples of a ray and sit in a handful of cells.  The
    // kernel is bound by the L1's miss path (one cache access per lane gather): only the first lane of
    // each run of equal cells fetches the 8 vertices, the others take them from it through the LDS crossbar.
    const bool dedup = (int)res <= dedup_max_res;  // wave-uniform
    const int lane = lane_id();
    for (int64_t tile = tm.tile0; tile * 256 < M; tile += tm.tstep) {
        const int64_t m = tile * 256 + threadIdx.x;
        const bool valid = m < M;  // (no early exit: the run logic below needs every lane of the wave)
        LevelPos p;
        p.gx = p.gy = p.gz = 0u; p.fx = p.fy = p.fz = 0.f;
        if (valid) p = level_pos(xyzs, m, bound, scale);
        uint32_t rows[8];
        corner_rows(p.gx, p.gy, p.gz, res, hsize, rows, meta.blocked);
        bool fetch = valid;
        int src = lane;
        if (dedup) {
            const RunInfo ri = wave_cell_runs(p.gx, p.gy, p.gz, valid, lane);
            src = ri.start;
            fetch = valid && ri.start == lane;
        }
        // issue the gathers first, blend afterwards (keeps up to 8 loads in flight per lane)
        CellRaw<TT> cell;
        cell.zero();
        if (fetch) {
            if (pair_loads && dense) {
#pragma unroll
                for (int c = 0; c < 8; c += 2) cell.load_pair(lt, rows[c], c);  // rows[c+1] == rows[c] + 1
            } else if (meta.blocked == 2) {
                // tiled: rows follow the dense index mod hsize (no x ^ h structure to pair loads on)
#pragma unroll
                for (int c = 0; c < 8; ++c) cell.load_one(lt, rows[c], c);
            } else if (pair_loads == 2 && CellRaw<TT>::kHasQuad && pow2 && (p.gx & 3u) != 3u) {
                // hashed, x mod 4 != 3: both x-neighbours sit in one aligned group of four rows (row = x ^ h: the group
                // is (x ^ h) & ~3) -- one 16-byte access instead of one 8-byte or two 4-byte ones
#pragma unroll
                for (int c = 0; c < 8; c += 2) cell.load_quad(lt, rows[c], rows[c + 1], c);
            } else if (pair_loads && pow2 && !(p.gx & 1u)) {
                // hashed, x even: the two x-neighbours are the two halves of one aligned pair
#pragma unroll
                for (int c = 0; c < 8; c += 2) {
                    cell.load_pair(lt, rows[c] & ~1u, c);
                    if (rows[c] & 1u) cell.swap_pair(c);
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) cell.load_one(lt, rows[c], c);
            }
        }
        if (dedup) cell.take_from_lane(src);
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t bx = c & 1, by = (c >> 1) & 1, bz = (c >> 2) & 1;
            const float wx = bx ? p.fx : 1.0f - p.fx;
            const float wy = by ? p.fy : 1.0f - p.fy;
            const float wz = bz ? p.fz : 1.0f - p.fz;
            const float w = (wx * wy) * wz;
            const float2 v = cell.get(c);
            a0 = fmaf(w, v.x, a0);
            a1 = fmaf(w, v.y, a1);
        }
        if (valid) Feat2<TO>::store(feat, (int64_t)l * level_stride + m, a0, a1);
    }
  }
}

// Backward, variant 0: one (sample, level) per thread, 16 global float atomics each.
template <typename TG>
__global__ void __launch_bounds__(256)
k_grid_backward_atomic(const float *__restrict__ xyzs, float bound, const TG *__restrict__ dfeat, GridMeta meta,
                       int64_t m_host, const int32_t *__restrict__ m_dev, int64_t level_stride,
                       float *__restrict__ dtable, int variant) {
    int64_t M = m_host;
    if (m_dev) { const int64_t md = *m_dev; M = md < M ? md : M; }
    const TileMap tm = tile_map(variant, meta.num_levels);
    if (!tm.ok) return;
    const int l = tm.level;
    const float scale = meta.scales[l];
    const uint32_t res = (uint32_t)meta.res[l];
    const uint32_t off = (uint32_t)meta.offsets[l];
    const uint32_t hsize = (uint32_t)(meta.offsets[l + 1] - meta.offsets[l]);
    float *lt = dtable + (int64_t)off * 2;
    for (int64_t tile = tm.tile0; tile * 256 < M; tile += tm.tstep) {
        const int64_t m = tile * 256 + threadIdx.x;
        if (m >= M) continue;
        const LevelPos p = level_pos(xyzs, m, bound, scale);
        const float2 gg = Feat2<TG>::load(dfeat + ((int64_t)l * level_stride + m) * 2, 0);
        uint32_t rows[8];
        corner_rows(p.gx, p.gy, p.gz, res, hsize, rows, meta.blocked);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t bx = c & 1, by = (c >> 1) & 1, bz = (c >> 2) & 1;
            const uint32_t row = rows[c];
            const float wx = bx ? p.fx : 1.0f - p.fx;
            const float wy = by ? p.fy : 1.0f - p.fy;
            const float wz = bz ? p.fz : 1.0f - p.fz;
            const float w = (wx * wy) * wz;
            atomicAdd(lt + (int64_t)row * 2, w * gg.x);
            atomicAdd(lt + (int64_t)row * 2 + 1, w * gg.y);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Backward, variants 2 / 3: two-pass bucketed scatter -- no global atomics anywhere.
//
// Scattered 8-byte float atomics run at the memory side at ~20 G requests/s chip-wide
// (MI355X_MICROARCH.md "Global float atomics"): 55 M vertex updates per frame cost ~10 ms that
// way.  Instead every level's table is cut into buckets of BK_ROWS consecutive rows (64 KiB of
// 64-bit accumulators = one LDS tile):
//   pass 1 (k_scatter_bin)    a work ITEM is 512 consecutive samples of one level.  One thread per sample computes its
//                             8 (row, w*g) records (runs of samples in one cell merged first on coarse levels) and the
//                             workgroup groups them by bucket in an LDS stage.  The stage IS the item's chunk of the
//                             record region: it is copied out as it stands (16 bytes per lane, perfectly coalesced),
//                             next to one table entry per (item, bucket) = (first slot, count) of the bucket's SEGMENT
//                             inside the chunk.  No reservations, no cursors, no head-room, no overflow path: an item
//                             owns ITEM_RECS = 4096 record slots, exactly what 512 samples can emit.
//   pass 2 (k_scatter_reduce) one workgroup per (bucket, slice) walks the items' segments of its bucket (the wave's
//                             lanes take consecutive records of the concatenated segments), accumulates them with
//                             64-bit fixed-point LDS atomics and finishes its 4096 rows (gradient add, bf16 output, or
//                             the fused Adam step).
// Every record has a fixed place that depends on the input only, every sum is an exact integer sum: the
// result is bitwise reproducible for ANY input (round 2's layout reserved spans with global atomics and fell back to
// float atomics when a bucket's region overflowed).
#ifndef LNERF_BK_SHIFT
#define LNERF_BK_SHIFT 12
#endif
constexpr int BK_SHIFT = LNERF_BK_SHIFT, BK_ROWS = 1 << BK_SHIFT;  // 4096 rows * 2 features * 8 B = 64 KiB of accumulators
// threads (= samples) per binning tile: template parameter BIN_T of k_scatter_bin (256 or 512)
constexpr int BK_MAX_PER_LEVEL = 256;                   // LDS counters per workgroup tile

// One scatter record.
//   Rec12 (variant 2): row inside the level + two f32 values: exact.
//   Rec8  (variant 3): row inside the BUCKET (12 bits; the bucket is implied by the region the record sits in)
//                      + the two values rounded (nearest-even) to 26-bit floats, sign + 8 exponent + 17 mantissa
//                      bits: relative rounding 2^-18 per addend instead of 2^-24.  One third less record traffic
//                      in both passes; meant for the bf16 configuration, whose gradients carry 2^-9 already.
struct Rec12 {
    uint32_t row;
    float v0, v1;
    static constexpr bool kPacked = false;
    static __device__ __forceinline__ Rec12 make(uint32_t row, float a, float b) {
        Rec12 r;
        r.row = row; r.v0 = a; r.v1 = b;
        return r;
    }
    __device__ __forceinline__ uint32_t row_in_bucket() const { return row & (uint32_t)(BK_ROWS - 1); }
    __device__ __forceinline__ float a() const { return v0; }
    __device__ __forceinline__ float b() const { return v1; }
};
// (native vector types: what __builtin_nontemporal_load / _store take)
typedef float nt_f4 __attribute__((ext_vector_type(4)));
typedef float nt_f2 __attribute__((ext_vector_type(2)));
typedef uint32_t nt_u2 __attribute__((ext_vector_type(2)));
typedef uint32_t nt_u4 __attribute__((ext_vector_type(4)));
// LNERF_BIN_NT (bit mask): non-temporal policy in the binning pass -- 1: dfeat loads (read once per step),
// 2: record stores (216 MB per frame: more than the Infinity Cache keeps until pass 2 reads them).  Measured together
// with LNERF_REDUCE_NT below, same box, three interleaved rounds (profiles/r03_exp_scatter.jsonl, steps Q / R):
// 2411 -> 2548 frames/s; bin 94.1 -> 86.5 us, reduce 124.5 -> 117.3, and the GATHER 78.5 -> 75.1 (its 24 MB table
// is no longer pushed out of the caches by the scatter's streams between two frames)
#ifndef LNERF_BIN_NT
#define LNERF_BIN_NT 3
#endif
struct alignas(8) Rec8 {
    uint32_t lo, hi;  // bits [0,12) row in bucket, [12,38) value 0, [38,64) value 1
    static constexpr bool kPacked = true;
    static __device__ __forceinline__ uint32_t f26(float v) {
        uint32_t u = __float_as_uint(v);
        if ((u & 0x7F800000u) != 0x7F800000u) u += 0x20u;  // finite: round to nearest, ties away from zero
        return u >> 6;
    }
    static __device__ __forceinline__ Rec8 make(uint32_t row, float a, float b) {
        const uint32_t qa = f26(a), qb = f26(b);
        Rec8 r;
        r.lo = (row & (uint32_t)(BK_ROWS - 1)) | (qa << 12);
        r.hi = (qa >> 20) | (qb << 6);
        return r;
    }
    __device__ __forceinline__ uint32_t row_in_bucket() const { return lo & (uint32_t)(BK_ROWS - 1); }
    __device__ __forceinline__ float a() const { return __uint_as_float((((lo >> 12) | (hi << 20)) & 0x3FFFFFFu) << 6); }
    __device__ __forceinline__ float b() const { return __uint_as_float((hi >> 6) << 6); }
};
static_assert(BK_SHIFT <= 12, "Rec8 stores 12 row bits");
// "This record is needed HERE, by every lane": an empty asm that reads the registers.  A load whose result is only used
// under a lane predicate is otherwise SUNK into the predicated block by the compiler -- one load, one s_waitcnt
// vmcnt(0), one use at a time instead of a batch of loads in flight (measured on the reduce pass: 2-3x its time).
__device__ __forceinline__ void pin_record(Rec8 &r) { asm volatile("" : "+v"(r.lo), "+v"(r.hi)); }
__device__ __forceinline__ void pin_record(Rec12 &r) { asm volatile("" : "+v"(r.row), "+v"(r.v0), "+v"(r.v1)); }

// Distance between two levels' maxima in uint32 words: one 128-byte line each (device-scope atomics that hit ONE line
// are served one after the other at the memory side, whatever words they name).
#ifndef LNERF_CUR_STRIDE
#define LNERF_CUR_STRIDE 32
#endif
constexpr int CUR_STRIDE = LNERF_CUR_STRIDE;
constexpr int ITEM_SAMPLES = 512;                 // samples per work item of pass 1 (= threads per workgroup)
constexpr int ITEM_RECS = ITEM_SAMPLES * 8;       // record slots of an item's chunk
// workspace header (bytes): [0, HDR_GMAX) level maxima (cleared before pass 1), then the item count of the last pass 1,
// then one record count per bucket (written by pass 2 for the finishing pass)
constexpr size_t HDR_GMAX_BYTES = (size_t)LNERF_MAX_LEVELS * CUR_STRIDE * sizeof(uint32_t);
constexpr size_t HDR_ITEMS_OFF = HDR_GMAX_BYTES;            // int32 [1] (+ padding to 128 bytes)
constexpr size_t HDR_ARRIVE_OFF = HDR_GMAX_BYTES + 128;     // int32 [9 x CUR_STRIDE]: arrival counters of the step's tail
                                                            // launch (root + 8 shards, a line each; zero between launches)
// slice arrival counters of pass 2, one per bucket, at a FIXED place whatever the level table (a process re-uses one
// workspace for every encoder: a region whose position depended on the bucket count would overlap another layout's
// record counts); zero in a fresh workspace (LNERF_SCATTER_ZERO_HEAD_BYTES), left zero by every call
constexpr size_t HDR_SLICE_ARRIVE_OFF = HDR_ARRIVE_OFF + 9 * CUR_STRIDE * sizeof(int32_t);
constexpr size_t HDR_BUCKETN_OFF = HDR_SLICE_ARRIVE_OFF + (size_t)LNERF_MAX_LEVELS * 256 * sizeof(int32_t);    // int32 [buckets]

struct BucketMeta {
    int nb[LNERF_MAX_LEVELS];            // buckets per level
    int bstart[LNERF_MAX_LEVELS + 1];    // first global bucket id of the level
    int slices[LNERF_MAX_LEVELS];        // pass-2 workgroups per bucket (worst case; the active count is decided on the device)
    int compact[LNERF_MAX_LEVELS];       // 1: merge runs of equal rows inside a wavefront before binning
    int wgstart[LNERF_MAX_LEVELS + 1];   // first pass-2 workgroup of the level
    int pstart[LNERF_MAX_LEVELS];        // sliced levels: first partial-sum tile of the level (pass 2 -> finish)
    int fstart[LNERF_MAX_LEVELS];        // sliced levels: first bucket index in the finishing pass's grid
    int n_items;                         // item capacity: ceil(m_host / ITEM_SAMPLES)
    int fix_bits;                        // exact 12-byte records: bits of the fixed-point addends (<= 44), chosen so that
                                         // m_host addends of the level's bound cannot overflow an int64 (see fix_scale)
    // chunk of (level l, item t): record slot ((int64)l * n_items + t) * ITEM_RECS;
    // segment table entry of (l, t, bucket b): ((int64)bstart[l] * n_items + (int64)t * nb[l] + b)
};

// sum of v over this lane's run, valid on the run's tail lane: difference of wave prefix sums
__device__ __forceinline__ float run_sum(float v, const RunInfo &r) {
    const float P = wave_inclusive_sum(v);
    const float Pm = __int_as_float(__builtin_amdgcn_ds_bpermute((r.start - 1) << 2, __float_as_int(P)));
    return r.start > 0 ? P - Pm : P;
}

// Phase stamps of the binning pass (diagnostic builds only: -DLNERF_STAMPS, tools/run_bin_stamps.sh).  Wave 0 of
// every workgroup drains its memory counters, reads the shader clock and adds the time since the previous stamp
// to a global per-phase total.
#ifdef LNERF_STAMPS
__device__ unsigned long long g_bin_stamps[16];
#define BIN_STAMP(k)                                                                              \
    do {                                                                                          \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                               \
        const unsigned long long now__ = __builtin_amdgcn_s_memtime();                            \
        stamp_acc__[k] += now__ - stamp_prev__;                                                   \
        stamp_prev__ = now__;                                                                     \
    } while (0)
#define BIN_STAMP_INIT()                                                                          \
    unsigned long long stamp_acc__[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                            \
    unsigned long long stamp_prev__ = __builtin_amdgcn_s_memtime()
#define BIN_STAMP_FLUSH()                                                                         \
    do {                                                                                          \
        if (threadIdx.x == 0)                                                                     \
            for (int k__ = 0; k__ < 10; ++k__) atomicAdd(&g_bin_stamps[k__], stamp_acc__[k__]);   \
    } while (0)
// per-workgroup log of pass 2: (entry, exit) on the constant 100 MHz clock + where it ran: 4 words per workgroup
__device__ unsigned long long g_wg_log[4 * 4096];
#define RED_STAMP(k) BIN_STAMP(k)
#define RED_STAMP_INIT()                                                                          \
    unsigned long long stamp_acc__[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};        \
    unsigned long long stamp_prev__ = __builtin_amdgcn_s_memtime()
#define RED_STAMP_FLUSH()                                                                         \
    do {                                                                                          \
        if (threadIdx.x == 0) {                                                                   \
            for (int k__ = 10; k__ < 14; ++k__) atomicAdd(&g_bin_stamps[k__], stamp_acc__[k__]);  \
            atomicAdd(&g_bin_stamps[15], 1ull);                                                   \
        }                                                                                         \
    } while (0)
#else
#define RED_STAMP(k) do { } while (0)
#define RED_STAMP_INIT() do { } while (0)
#define RED_STAMP_FLUSH() do { } while (0)
#define BIN_STAMP(k) do { } while (0)
#define BIN_STAMP_INIT() do { } while (0)
#define BIN_STAMP_FLUSH() do { } while (0)
#endif

// ---- pass 1: k_scatter_bin ---------------------------------------------------------------------------------------
// A work item is BIN_T = 512 consecutive samples of ONE level; PERSISTENT workgroups (3 per CU) stride over the
// tile-major (tile, level) list, the level rotated by one per round, and fetch the next item's inputs while the current
// one is processed.  Per item: cell, rows, runs -> every record ranked inside its bucket with a returning LDS counter
// -> (the values w * g, run sums on coarse levels, are computed behind those atomics) -> barrier -> count scan (every
// wave computes it: no idle waves, no extra barrier) -> records written to their slot of the LDS stage -> barrier ->
// the stage copied out as the item's chunk, 16 bytes per lane, and the (first slot, count) of every bucket's segment
// written to the segment table.  Nothing in the item waits for a global round trip: the only global accesses are the
// prefetch of the next item's inputs and the two coalesced stores at the end.
// The level's largest |value| (fixed-point scale of pass 2) is bounded from |g| (weights <= 1, runs <= 64 samples),
// one LDS maximum per level and workgroup; records are packed with bit-field inserts; run sums use fused DPP adds.
constexpr int BIN_T = ITEM_SAMPLES;        // threads per workgroup = samples per item

// fast f32 -> 26-bit float (round to nearest, ties away from zero: one add on the sign-magnitude bits; symmetric in
// the sign, and a tie is one value in 64), valid for finite values
__device__ __forceinline__ uint32_t f26_round(float v) {
    return __float_as_uint(v) + 0x20u;  // (low 6 bits are dropped by the packing)
}
template <typename REC, bool CAREFUL> struct PackRec;
template <bool CAREFUL> struct PackRec<Rec12, CAREFUL> {
    static __device__ __forceinline__ Rec12 make(uint32_t row, float a, float b) { return Rec12::make(row, a, b); }
};
template <> struct PackRec<Rec8, true> {   // non-finite values present in the wavefront: the reference packing
    static __device__ __forceinline__ Rec8 make(uint32_t row, float a, float b) { return Rec8::make(row, a, b); }
};
template <> struct PackRec<Rec8, false> {  // bits [0,12) row, [12,38) value 0, [38,64) value 1 -- same layout, fewer ops
    static __device__ __forceinline__ Rec8 make(uint32_t row, float a, float b) {
        const uint32_t ua = f26_round(a), ub = f26_round(b);
        Rec8 r;
        r.lo = ((ua << 6) & 0xFFFFF000u) | (row & 0xFFFu);
        r.hi = (ua >> 26) | (ub & 0xFFFFFFC0u);
        return r;
    }
};

// maximum over the wave of unsigned values, returned in every lane: one fused DPP max per step (a dependent chain:
// every DPP read needs the two wait states after the VALU write, which the compiler cannot see inside inline asm)
__device__ __forceinline__ unsigned int wave_max_u32(unsigned int v) {
    asm volatile("s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_u32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(v));
    return (unsigned int)__builtin_amdgcn_readlane((int)v, 63);
}

// Sums over RUNS of lanes (RunInfo), valid on every lane as the sum from the run's first lane up to the lane itself:
// a segmented Hillis-Steele scan, one fused DPP multiply-add per value and step -- the addend of a lane whose source
// lies before its run's first lane is multiplied by 0.  Unlike "wave prefix sum minus the prefix before the run" it
// needs no lane permutes, no subtraction (and has none of its cancellation), and a wave whose longest run is short
// skips the long-distance steps: all conditions are wave-uniform scalar tests on the run-head mask.
__device__ __forceinline__ void wave_run_sums_x16(float (&a)[8], float (&b)[8], const RunInfo &r, int lane) {
    const int d = lane - r.start;  // lanes of the run before this one
    const unsigned long long H = r.heads;
    const unsigned long long H2 = H | (H << 1), H4 = H2 | (H2 << 2), H8 = H4 | (H4 << 4);
#define LNERF_SEG_STEP(ctrl, cond)                                                                                  \
    {                                                                                                               \
        const float f = (cond) ? 1.0f : 0.0f;                                                                       \
        asm volatile("s_nop 1" ::: );                                                                               \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                                                             \
            asm volatile("v_fmac_f32_dpp %0, %0, %1 " ctrl : "+v"(a[i]) : "v"(f));                                  \
            asm volatile("v_fmac_f32_dpp %0, %0, %1 " ctrl : "+v"(b[i]) : "v"(f));                                  \
        }                                                                                                           \
        asm volatile("s_nop 1" ::: );                                                                               \
    }
    if (H != ~0ull) LNERF_SEG_STEP("row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0", d >= 1)
    if (H2 != ~0ull) LNERF_SEG_STEP("row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0", d >= 2)
    if (H4 != ~0ull) LNERF_SEG_STEP("row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0", d >= 4)
    if (H8 != ~0ull) LNERF_SEG_STEP("row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0", d >= 8)
    // runs that continue over a row of 16 lanes: the previous row's last lane holds the run's sum so far
    if ((H & 0x0001000000010000ull) != 0x0001000000010000ull)
        LNERF_SEG_STEP("row_bcast:15 row_mask:0xa bank_mask:0xf", d > (lane & 15))
    if (!((H >> 32) & 1ull)) LNERF_SEG_STEP("row_bcast:31 row_mask:0xc bank_mask:0xf", r.start < 32)
#undef LNERF_SEG_STEP
}

// what the binning pass needs to know about a level (read from the kernel arguments in the kernel body only: the
// lambdas below take it by value, so the argument structs are never copied to scratch)
struct BinLevel {
    float scale;
    uint32_t res, hsize;
    int level, nb, b0;
    bool compact;
};
#define LNERF_BIN_LEVEL(lv)                                                                                          \
    BinLevel {                                                                                                       \
        meta.scales[lv], (uint32_t)meta.res[lv], (uint32_t)(meta.offsets[(lv) + 1] - meta.offsets[lv]), (lv),        \
            bm.nb[lv], bm.bstart[lv], bm.compact[lv] != 0                                                            \
    }

template <typename REC>
__global__ void __launch_bounds__(BIN_T, (sizeof(REC) == 8 ? 6 : 4))
k_scatter_bin(const float *__restrict__ xyzs, float bound, const float *__restrict__ dfeat, GridMeta meta, BucketMeta bm,
              int64_t m_host, const int32_t *__restrict__ m_dev, int64_t level_stride, unsigned int *__restrict__ gmax,
              int32_t *__restrict__ items_out, uint32_t *__restrict__ segtab, REC *__restrict__ recs, int skip_zero,
              int lv_lo, int lv_hi) {
    __shared__ int s_cnt[2][BK_MAX_PER_LEVEL];  // records of the item per bucket (two sets: items alternate)
    __shared__ int s_off[BK_MAX_PER_LEVEL];     // first slot of the bucket's segment in the stage (= in the chunk)
    __shared__ __attribute__((aligned(16))) REC s_stage[ITEM_RECS];  // the item's chunk (48 KiB, 32 KiB packed)
    __shared__ unsigned int s_lmax[LNERF_MAX_LEVELS];         // per level: bound of |value| seen by this workgroup
    int32_t M = (int32_t)m_host;
    if (m_dev) { const int32_t md = *m_dev; M = md < M ? md : M; }
    const int L = lv_hi - lv_lo;   // levels of this launch: [lv_lo, lv_hi)
    const int tid = threadIdx.x, lane = tid & 63;
    // item k of this workgroup: tile t0 + k * tstep, level lv_lo + (l0 + k) mod L   (gridDim.x is a multiple of L)
    const int tstep = gridDim.x / L;
    const float two_b = 2.0f * bound;
    const bool pow2_bound = (__float_as_uint(two_b) & 0x007FFFFFu) == 0u;
    // (exact when the bound is a power of two, the only case it is used in; wave-uniform, kept in a scalar register)
    float inv_two_b;
    asm("v_readfirstlane_b32 %0, %1" : "=s"(inv_two_b) : "v"(1.0f / two_b));
    for (int i = tid; i < 2 * BK_MAX_PER_LEVEL; i += BIN_T) (&s_cnt[0][0])[i] = 0;
    if (tid < LNERF_MAX_LEVELS) s_lmax[tid] = 0u;
    if (blockIdx.x == 0 && tid == 0) *items_out = (M + BIN_T - 1) / BIN_T;  // pass 2 walks exactly these items
    // ---- inputs of an item (5 dwords per lane), fetched while the previous item is processed
    float n_x = 0.f, n_y = 0.f, n_z = 0.f;
    float2 n_g = make_float2(0.f, 0.f);
    auto fetch = [&](int lv, int tl) __attribute__((always_inline)) {
        const int mm = tl * BIN_T + tid;
        n_x = n_y = n_z = 0.f;
        n_g = make_float2(0.f, 0.f);
        if (mm < M) {
            const float2 *gp = reinterpret_cast<const float2 *>(dfeat) + ((int64_t)lv * level_stride + mm);
            if (LNERF_BIN_NT & 1) {   // (read once per step: keep it out of the caches the table and the records use)
                const nt_f2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f2 *>(gp));
                n_g = make_float2(v.x, v.y);
            } else {
                n_g = *gp;
            }
            n_x = xyzs[(int64_t)mm * 3]; n_y = xyzs[(int64_t)mm * 3 + 1]; n_z = xyzs[(int64_t)mm * 3 + 2];
        }
    };
    int l = lv_lo + (int)(blockIdx.x % L);
    int tile = (int)(blockIdx.x / L);
    bool have = tile * BIN_T < M;
    if (have) fetch(l, tile);
    BIN_STAMP_INIT();
    __syncthreads();
    int hk = 0;  // items so far (selects the counter set)
    while (have) {
        const BinLevel lv = LNERF_BIN_LEVEL(l);
        const int nb = lv.nb;
        const int l_next = l + 1 == lv_hi ? lv_lo : l + 1;
        const int tile_next = tile + tstep;
        const bool have_next = tile_next * BIN_T < M;
        BIN_STAMP(0);
        // ---- A: cell, rows, runs, and WHICH lanes append records.  Samples behind a ray's termination point
        // (T < T_thresh) get dsigma = drgb = 0 from the compositing backward, hence dfeat = 0 exactly: a run (or
        // sample) whose gradients are all zero appends nothing, and a wavefront of 64 such samples skips its
        // arithmetic altogether.
        const int m = tile * BIN_T + tid;
        const bool valid = m < M;
        const float2 gg = n_g;
        const bool nzg = valid && (gg.x != 0.f || gg.y != 0.f);
        const unsigned long long nzmask = __ballot(nzg);
        const bool wave_live = !skip_zero || nzmask != 0ull;
        LevelPos p;
        RunInfo ri;
        ri.start = lane; ri.tail = true; ri.heads = ~0ull;
        uint32_t row[8];
        bool emit = false;
        if (wave_live) {
            {   // (lanes past the end hold zeros from the fetch: same arithmetic, nothing emitted)
                float px = n_x + bound, py = n_y + bound, pz = n_z + bound;
                if (pow2_bound) { px *= inv_two_b; py *= inv_two_b; pz *= inv_two_b; }   // == the division, exactly
                else { px /= two_b; py /= two_b; pz /= two_b; }
                px = px * lv.scale; py = py * lv.scale; pz = pz * lv.scale;
                px = px + 0.5f; py = py + 0.5f; pz = pz + 0.5f;
                const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
                p.gx = (uint32_t)(int)flx; p.gy = (uint32_t)(int)fly; p.gz = (uint32_t)(int)flz;
                p.fx = px - flx; p.fy = py - fly; p.fz = pz - flz;
            }
            corner_rows(p.gx, p.gy, p.gz, lv.res, lv.hsize, row, meta.blocked);
            if (lv.compact) {  // wave-uniform: coarse level, merge runs of samples in the same cell first
                // (the lane number is made opaque per item: the 64-bit lane masks derived from it are cheaper to
                // recompute than to keep -- hoisted out of the item loop they were spilled to scratch)
                int lane_v = lane;
                asm volatile("" : "+v"(lane_v));
                ri = wave_cell_runs(p.gx, p.gy, p.gz, valid, lane_v);
                const unsigned long long seg = (nzmask >> ri.start) & ((2ull << (lane_v - ri.start)) - 1ull);
                emit = valid && ri.tail && (!skip_zero || seg != 0ull);
            } else {
                emit = valid && (!skip_zero || nzg);
            }
        } else {  // nothing is emitted: rows and position are never looked at (defined without an instruction)
#pragma unroll
            for (int c = 0; c < 8; ++c) asm("" : "=v"(row[c]));
            asm("" : "=v"(p.gx), "=v"(p.gy), "=v"(p.gz), "=v"(p.fx), "=v"(p.fy), "=v"(p.fz));
        }
        // ---- D (a lambda: placed behind the ranking atomics): the values w * g (run sums on coarse levels), packed
        // into records; the bound of |value| goes to the level's LDS maximum
        REC rec[8];
        auto values = [&]() __attribute__((always_inline)) {
            if (!wave_live) return;
            float v0[8], v1[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float wx = (c & 1) ? p.fx : 1.0f - p.fx;
                const float wy = (c & 2) ? p.fy : 1.0f - p.fy;
                const float wz = (c & 4) ? p.fz : 1.0f - p.fz;
                const float w = (wx * wy) * wz;
                v0[c] = w * gg.x;
                v1[c] = w * gg.y;
            }
            float mx = fmaxf(fabsf(gg.x), fabsf(gg.y));  // weights are <= 1 ...
            const bool odd = ((__float_as_uint(gg.x) & 0x7F800000u) == 0x7F800000u) ||
                             ((__float_as_uint(gg.y) & 0x7F800000u) == 0x7F800000u);  // NaN / inf in the gradient
            if (lv.compact) {
                wave_run_sums_x16(v0, v1, ri, lane);          // the run's tail lane holds the run sum
                mx *= 64.0f;                             // ... and a run sums at most 64 samples
            }
            const bool any_odd = __ballot(odd) != 0ull;
            // (a per-lane LDS maximum, filtered by the current bound, measured 32 us SLOWER than this wave reduction)
            const unsigned int mb = wave_max_u32(__float_as_uint(any_odd ? 3.0e38f : mx));  // (bits of floats >= 0 order as uints)
            if (lane == 0 && mb != 0u) atomicMax(&s_lmax[lv.level], mb);
            if (!any_odd) {
#pragma unroll
                for (int c = 0; c < 8; ++c) rec[c] = PackRec<REC, false>::make(row[c] & ((1u << 20) - 1u), v0[c], v1[c]);
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) rec[c] = PackRec<REC, true>::make(row[c] & ((1u << 20) - 1u), v0[c], v1[c]);
            }
        };
        // ranks the wavefront's records of corner c in `counters` (LDS): where the lanes of a wave mostly target one or
        // two buckets (tiny tables) one LDS atomic per (wave, bucket) instead of one per lane -- same-address LDS
        // atomics serialise
        auto rank_by_ballot = [&](int *counters, int c) __attribute__((always_inline)) {
            const int b = (int)(row[c] >> BK_SHIFT);
            int rk = 0;
            unsigned long long todo = __ballot(emit);
            while (todo) {
                const int leader = __builtin_amdgcn_readfirstlane(__ffsll((long long)todo) - 1);
                const int bl = __builtin_amdgcn_readlane(b, leader);
                const unsigned long long mm = __ballot(emit && b == bl);
                int base = 0;
                if (lane == leader) base = atomicAdd(&counters[bl], __popcll(mm));
                base = __builtin_amdgcn_readlane(base, leader);
                if (emit && b == bl) rk = base + mbcnt(mm);
                todo &= ~mm;
            }
            row[c] |= (uint32_t)rk << 20;
        };
        BIN_STAMP(1);
        const int cur = hk & 1;
        ++hk;
        // ---- B: the rank of a record inside its bucket, among the item's records (< 4096), is kept in bits [20, 32) of
        // its row (rows of a level are < 2^20: at most 256 buckets of 4096 rows)
        constexpr uint32_t ROW_MASK = (1u << 20) - 1u;
        if (nb <= 32 && !lv.compact) {  // wave-uniform: every lane emits into one or two buckets
#pragma unroll
            for (int c = 0; c < 8; ++c) rank_by_ballot(s_cnt[cur], c);
        } else if (emit) {
#pragma unroll
            for (int c = 0; c < 8; ++c) row[c] |= (uint32_t)atomicAdd(&s_cnt[cur][row[c] >> BK_SHIFT], 1) << 20;
        }
        BIN_STAMP(2);
        // the next item's inputs go into the memory queue now; they are consumed at the top of the next iteration
        if (have_next) fetch(l_next, tile_next);
        values();
        BIN_STAMP(4);
        __syncthreads();  // barrier 1: the item's bucket counts are final
        BIN_STAMP(3);
        // (the other set was last read behind barrier 2 of the previous item: clear it for the next one)
        for (int i = tid; i < BK_MAX_PER_LEVEL; i += BIN_T) s_cnt[cur ^ 1][i] = 0;
        // ---- E: exclusive scan of the bucket counts.  EVERY wave computes it (4 buckets per lane, one DPP scan) and
        // writes the same offsets: a wave reads s_off only after its own writes, so no barrier and no idle waves
        int total;
        {
            int c4[4], sum = 0;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int i = lane * 4 + kk;
                c4[kk] = i < nb ? s_cnt[cur][i] : 0;
                sum += c4[kk];
            }
            const int inc = wave_inclusive_sum_i(sum);
            int run = inc - sum;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int i = lane * 4 + kk;
                if (i < nb) s_off[i] = run;
                run += c4[kk];
            }
            total = __builtin_amdgcn_readlane(inc, 63);
        }
        // ---- F: the records into their slot of the stage = of the chunk
        if (emit) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int b = (int)((row[c] & ROW_MASK) >> BK_SHIFT);
                s_stage[s_off[b] + (int)(row[c] >> 20)] = rec[c];
            }
        }
        BIN_STAMP(5);
        __syncthreads();  // barrier 2: the stage is complete
        BIN_STAMP(7);
        // the prefetched inputs are pinned in registers here, so that the next item starts without waiting for the
        // stores below to be acknowledged (one in-order memory counter covers loads and stores)
        asm volatile("" : "+v"(n_g.x), "+v"(n_g.y), "+v"(n_x), "+v"(n_y), "+v"(n_z));
        // ---- G: the chunk, 16 bytes per lane (a chunk starts on a multiple of 16 bytes; the last unit may carry one
        // stale record behind the item's last one: never read), and the segment table entries of the item
        {
            uint4 *dst = reinterpret_cast<uint4 *>(recs + ((int64_t)lv.level * bm.n_items + tile) * ITEM_RECS);
            const uint4 *srcq = reinterpret_cast<const uint4 *>(s_stage);
            const int n16 = (total * (int)sizeof(REC) + 15) >> 4;
            for (int i = tid; i < n16; i += BIN_T) {
                if (LNERF_BIN_NT & 2) {
                    const uint4 q = srcq[i];
                    nt_u4 v = {q.x, q.y, q.z, q.w};
                    __builtin_nontemporal_store(v, reinterpret_cast<nt_u4 *>(dst + i));
                } else {
                    dst[i] = srcq[i];
                }
            }
            if (tid < nb)
                segtab[(int64_t)lv.b0 * bm.n_items + (int64_t)tile * nb + tid] =
                    (uint32_t)s_off[tid] | ((uint32_t)s_cnt[cur][tid] << 16);
        }
        BIN_STAMP(8);
        // (the next item rewrites s_off / s_stage only behind ITS barrier 1, which every wave reaches after finishing
        // the copy above; it clears this item's counter set behind that barrier too)
        l = l_next; tile = tile_next; have = have_next;
    }
    __syncthreads();
    if (tid >= lv_lo && tid < lv_hi && s_lmax[tid] != 0u) atomicMax(&gmax[tid * CUR_STRIDE], s_lmax[tid]);  // one value per LEVEL and workgroup
    BIN_STAMP_FLUSH();
}
#undef LNERF_BIN_LEVEL

// Pass 2.  LDS float atomics run at ~0.5 lane/clk on gfx950 while integer LDS atomics run at the
// plain-store rate (measured: profiles/README.md, "reduce_dbg"), so the tile accumulates in 64-bit
// FIXED POINT: every value is scaled by a power of two chosen from the level's bound of |value|
// (found by pass 1) so that |q| < 2^44 (12-byte records; 2^30 with the 8-byte ones, see fix_scale), which leaves
// 2^19 additions of head-room in an int64.  The scaling is exact, the integer sum is exact and
// order-independent, and the only rounding is the quantisation of each addend to 2^-45 (2^-31) of the
// level's bound plus one final conversion to f32: the result is bitwise reproducible; with the 12-byte
// records it is at least as accurate as an f32 running sum.
// records per slice workgroup of pass 2 (a bucket with fewer records is reduced by one workgroup)
constexpr int REDUCE_SLICE_RECS = 16384;

// Optional fused table update (lnerf_grid_encode_backward_adam): where pass 2 (or its finishing kernel) owns a
// row outright it applies the Adam step straight from the fixed-point sum: the gradient of the table never travels
// through HBM (42 -> 26 bytes per table entry and step).
struct FusedUpdate {
    float *p, *m, *v;
    uint16_t *shadow;    // optional bf16 copy of p, refreshed in the same pass
    AdamArgs a;
    uint16_t *grad_out;  // when set: no Adam step -- every row's finished sum is WRITTEN as bf16 here instead (the wire
                         // format of the data-parallel all-reduce: no zero fill, no read-modify-write, no cast)
};

// power-of-two scale of a level's fixed-point sums: |value| < 2^(e-126) (e = biased exponent of the level's bound
// found by pass 1) is scaled by 2^(BITS+126-e), which puts every addend below 2^BITS; split so that both factors are
// normal floats.  BITS = 44 (exact 12-byte records: quantum 2^-44 of the bound, 2^19 additions of head-room in an
// int64, conversion through the 64-bit software path) or 30 (8-byte records, whose values carry 17 mantissa bits
// anyway: quantum 2^-30 of the bound, conversion with the native v_cvt_i32_f32, a third of the pass's vector work).
struct FixScale {
    float sc_a, sc_b, un_a, un_b;
};
__device__ __forceinline__ FixScale fix_scale(unsigned int gmax_bits, int bits) {
    int e = (int)(gmax_bits >> 23);
    e = e < 1 ? 1 : (e > 254 ? 254 : e);
    int k = bits + 126 - e;
    k = k > 200 ? 200 : k;
    FixScale f;
    f.sc_a = ldexpf(1.0f, k / 2); f.sc_b = ldexpf(1.0f, k - k / 2);
    f.un_a = ldexpf(1.0f, -(k / 2)); f.un_b = ldexpf(1.0f, -(k - k / 2));
    return f;
}
template <typename REC> struct FixBits { static constexpr int kBits = REC::kPacked ? 30 : 44; };
template <int BITS> __device__ __forceinline__ long long to_fixed(float x);
template <> __device__ __forceinline__ long long to_fixed<44>(float x) { return __float2ll_rn(x); }
template <> __device__ __forceinline__ long long to_fixed<30>(float x) { return (long long)__float2int_rn(x); }  // |x| < 2^30
// slices a bucket with n records is cut into (decided on the device from the actual count; the launch provides
// `smax` workgroups per bucket for the worst case)
__device__ __forceinline__ int active_slices(int n, int smax) {
    int S = (n + REDUCE_SLICE_RECS - 1) / REDUCE_SLICE_RECS;
    return S < 1 ? 1 : (S > smax ? smax : S);
}

// A bucket summed by ONE workgroup: the workgroup adds its tile to dtable (or applies the Adam step, FUSE).
// A bucket cut into slices (few, heavily loaded coarse buckets): every slice stores its EXACT 64-bit partial sums
// as a tile of `partials`, and the slice that arrives last adds the tiles up -- integer addition, so the result does not depend
// on how many slices there were or in which order they ran: the whole gradient is bitwise reproducible.
//
// The records of bucket b are the segments (first slot, count) = segtab entry of (item, b), one per item of pass 1, inside
// the items' chunks.  Wave w of the workgroup takes items w, w + 16, ...: it reads 64 of its entries with one load and
// walks the concatenation of those segments 64 records per round (see the loop).  LNERF_REDUCE_ROUNDS rounds of loads
// are in flight per lane.  Built and measured on the way (profiles/r03_exp_scatter.jsonl): the segment of a lane found by
// a binary search through ds_bpermute (+12 us: the permutes share the LDS pipe with the atomics); one segment per round
// (half-empty waves: three times the instructions, 2-3x the time).
#ifndef LNERF_REDUCE_ROUNDS
#define LNERF_REDUCE_ROUNDS 8
#endif
#ifndef LNERF_REDUCE_XCD
#define LNERF_REDUCE_XCD 1
#endif
// how a lane of the record loop finds its record: 0 = scalar walk over the segments a round spans, 1 = start bitmap
#ifndef LNERF_REDUCE_WALK
#define LNERF_REDUCE_WALK 1
#endif
// ---- the step's TAIL: what is left of a single-GPU step besides the scatter.  In a replayed graph a dependent dispatch
// costs ~4.5 us whatever it computes, and three of them sat behind pass 2 for a few microseconds of work: the finishing
// pass of the sliced buckets (pass 2 does it itself now), the sum of the MLP's gradient slabs and the Adam step of the
// small parameters.  One SLAB BLOCK (256 threads) takes 16 parameters of the MLP: it sums their column of the gradient
// slabs in a fixed order (k_mlp_reduce_slabs' arithmetic: deterministic) and applies the Adam step straight from the sum
// -- the weight gradients never exist in memory; updated weights are mirrored into the bf16 weight fragments
// (lnerf_mlp_fragment_maps).  The LAST block of a launch to arrive advances the device step counter and leaves the
// scatter's level maxima zero for the next step (every other block has read both by then, and the arrival is a
// device-scope atomic).  The slab blocks run
//   * as EXTRA workgroups of pass 2 itself (lnerf_grid_encode_backward_adam_tail: four slab blocks per 1024-thread
//     workgroup, in front of the buckets; the step then has no launch behind pass 2 at all), or
//   * as their own launch (lnerf_step_tail: k_step_tail), where other small parameters must be stepped first.
struct SlabAdam {
    const float *slabs;
    int n_slabs, out_dim;
    float *p[6], *m[6], *v[6];        // w1, b1, w2, b2, w3, b3
    const int32_t *map[3];            // optional: fragment positions of w1, w2, w3 (two per weight)
    uint16_t *shadow;                 // the bf16 fragment image the maps point into
    float lr;
};
constexpr int TAIL_P = 16, TAIL_G = 16;   // parameters per block, slab groups (as k_mlp_reduce_slabs)

constexpr int TAIL_SLABS_PER_LANE = MLP_BWD_MAX_BLOCKS / TAIL_G;   // 32: every slab load of a lane in flight at once
constexpr int TAIL_SHARDS = 8;                                     // arrival counters (one 128-byte line each)

// One slab block: `blk` = index of the block of 16 parameters, `lt` = thread inside the block (0..255), `part` = its
// [TAIL_G][TAIL_P] floats of LDS.  `a`: the table's Adam arguments; `step_now` the device step counter when
// a.step_dev is set (requested by the caller with ONE device-scope atomic load per wave -- the last block of the launch
// to arrive rewrites it).  Every load that depends on nothing is requested first and together: the launch is a handful
// of dependent round trips per block.  Contains block-wide barriers at named-barrier-free places: call it with all 256
// threads of the block (the 1024-thread form synchronises the whole workgroup, see the caller).
// BATCH: slab values a lane keeps in flight (32 = all of them: the stand-alone launch; 8 inside pass 2, whose 64
// registers per lane must not spill -- the sum takes the slabs in the same order either way).
template <int BATCH, typename SYNC>
__device__ __forceinline__ void slab_block(int blk, int lt, const SlabAdam &sa, AdamArgs a, int32_t step_now,
                                           float (*part)[TAIL_P], SYNC sync) {
    static_assert(TAIL_SLABS_PER_LANE % BATCH == 0, "whole batches");
    const int pi = lt & (TAIL_P - 1), sg = lt / TAIL_P;
    const int p = blk * TAIL_P + pi;
    int k = -1, i = 0;   // slab column -> (tensor, element)
    float Pw = 0.f, Mw = 0.f, Vw = 0.f;
    int2 at = make_int2(-1, -1);
    if (p < MLP_SLAB) {
        if (p < MLP_SL_B1) { k = 0; i = p - MLP_SL_W1; }
        else if (p < MLP_SL_W2) { k = 1; i = p - MLP_SL_B1; }
        else if (p < MLP_SL_B2) { k = 2; i = p - MLP_SL_W2; }
        else if (p < MLP_SL_W3) { k = 3; i = p - MLP_SL_B2; }
        else if (p < MLP_SL_B3) { if ((p - MLP_SL_W3) / MLP_HID < sa.out_dim) { k = 4; i = p - MLP_SL_W3; } }
        else { if (p - MLP_SL_B3 < sa.out_dim) { k = 5; i = p - MLP_SL_B3; } }
    }
    float vsl[BATCH];
    auto fetch = [&](int j0) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < BATCH; ++j) {
            const int bsl = sg + TAIL_G * (j0 + j);
            vsl[j] = (p < MLP_SLAB && bsl < sa.n_slabs) ? sa.slabs[(int64_t)bsl * MLP_SLAB + p] : 0.f;
        }
    };
    fetch(0);
    if (sg == 0 && k >= 0) {
        Pw = sa.p[k][i]; Mw = sa.m[k][i]; Vw = sa.v[k][i];
        if (sa.shadow && !(k & 1)) at = reinterpret_cast<const int2 *>(sa.map[k >> 1])[i];
    }
    if (a.step_dev) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // (the counter has been READ: see the arrival)
        adam_bias_at(a, step_now);
    } else {
        adam_bias(a);
    }
    a.zero_grad = 0;
    float sum = 0.f;   // (k_mlp_reduce_slabs' order: slabs sg, sg + 16, ...)
    for (int j0 = 0;; j0 += BATCH) {
#pragma unroll
        for (int j = 0; j < BATCH; ++j) sum += vsl[j];
        if (j0 + BATCH >= TAIL_SLABS_PER_LANE) break;
        fetch(j0 + BATCH);
    }
    part[sg][pi] = sum;
    sync();
    if (sg == 0 && k >= 0) {
        sum = part[0][pi];
#pragma unroll
        for (int g = 1; g < TAIL_G; ++g) sum += part[g][pi];
        AdamArgs am = a;
        am.lr = sa.lr;
        adam_one(Pw, sum, Mw, Vw, am);
        sa.p[k][i] = Pw; sa.m[k][i] = Mw; sa.v[k][i] = Vw;
        const uint16_t h = f32_to_bf16(Pw);   // weights (k = 0, 2, 4) are mirrored into their two fragment positions
        if (at.x >= 0) sa.shadow[at.x] = h;
        if (at.y >= 0) sa.shadow[at.y] = h;
    }
}

// arrival of a block (call from ONE thread, after the block's reads of the step counter and the level maxima have
// returned), two levels: 8 shard counters (a line each: ~600 arrivals on ONE word queue for 7 us at the memory side), the
// block that completes a shard arrives at the root, the block that completes the root is the last of the launch
__device__ __forceinline__ void tail_arrive(int32_t *arrive, int total, int block, int32_t *tick, int32_t step_now,
                                            unsigned int *gmax, int do_tick, int clear_gmax) {
    const int sh = block & (TAIL_SHARDS - 1);
    const int mine = (total - sh + TAIL_SHARDS - 1) / TAIL_SHARDS;   // blocks of this shard
    int32_t *cnt = arrive + (1 + sh) * CUR_STRIDE;
    if (__hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == mine - 1) {
        __hip_atomic_store(cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int shards = total < TAIL_SHARDS ? total : TAIL_SHARDS;
        if (__hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1) {
            __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (do_tick) __hip_atomic_store(&tick[0], step_now + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (clear_gmax)
                for (int l = 0; l < LNERF_MAX_LEVELS; ++l)
                    __hip_atomic_store(&gmax[l * CUR_STRIDE], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// what pass 2 does besides the buckets when it closes the step (all zero: nothing)
struct TailJob {
    SlabAdam sa;
    int blocks;            // leading workgroups of the launch that run slab blocks (four each)
    int32_t *tick;         // device step counter pair
    int32_t *arrive;       // arrival counters (workspace header)
    int do_tick, clear_gmax;
    int rev_lo, rev_hi;    // work units [rev_lo, rev_hi) are taken in DESCENDING order (0, 0: none)
};

// LNERF_REDUCE_NT (bit mask): non-temporal policy on the once-per-step streams of the reduce pass -- 1: parameter /
// moment loads, 2: their stores (the bf16 shadow the gather reads keeps the default policy), 4: the record loads
// (measured: the gather gains 1.5 us more, the reduce pass loses 8 -- off)
#ifndef LNERF_REDUCE_NT
#define LNERF_REDUCE_NT 3
#endif
__device__ __forceinline__ float4 ld_f4(const float4 *p) {
    if (LNERF_REDUCE_NT & 1) {
        const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4 *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}
__device__ __forceinline__ void st_f4(float4 *p, const float4 &x) {
    if (LNERF_REDUCE_NT & 2) {
        nt_f4 v = {x.x, x.y, x.z, x.w};
        __builtin_nontemporal_store(v, reinterpret_cast<nt_f4 *>(p));
    } else {
        *p = x;
    }
}
template <typename REC> __device__ __forceinline__ REC ld_rec(const REC *p) { return *p; }
template <> __device__ __forceinline__ Rec8 ld_rec<Rec8>(const Rec8 *p) {
    if (LNERF_REDUCE_NT & 4) {
        const nt_u2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u2 *>(p));
        Rec8 r;
        r.lo = v.x; r.hi = v.y;
        return r;
    }
    return *p;
}

template <int RT, typename REC, bool FUSE>
__device__ __forceinline__ void scatter_reduce_one(int wg, const GridMeta &meta, const BucketMeta &bm,
                                                   const int32_t *__restrict__ items_dev,
                                                   const uint32_t *__restrict__ segtab, int32_t *__restrict__ bucket_n,
                                                   int32_t *__restrict__ slice_arrive,
                                                   unsigned int *gmax, const REC *__restrict__ recs,
                                                   float *__restrict__ dtable, long long *__restrict__ partials,
                                                   const FusedUpdate &fu, const SlabAdam &sa, int32_t step_now,
                                                   bool have_step) {
    // (declared HERE, not passed in: a pointer parameter loses the LDS address space and every ds_add_u64 becomes a
    // flat atomic -- measured 0.211 -> 0.275 ms for the scatter call)
    __shared__ long long acc[BK_ROWS * 2];  // [feature][row]: a wave's 64 random rows spread over 32 bank pairs
    __shared__ int s_red[RT / 64];
#if LNERF_REDUCE_WALK
    __shared__ uint32_t s_bmp[RT / 64][128];            // per wave: segment-start bitmap of a sub-batch (4096 records)
    __shared__ uint32_t s_soff[RT / 64][64];            // per wave: (chunk slot - flat start) of its non-empty segments
#endif
    constexpr int NW = RT / 64;
    if (wg < 0) {
        // a SLAB workgroup of the closing launch (wg = -1 - index): RT / 256 slab blocks, their 1 KiB of LDS each carved
        // out of the accumulator tile (used HERE, through the array itself: see above)
        const int sub = (int)threadIdx.x >> 8, lt = (int)threadIdx.x & 255;
        float (*part)[TAIL_P] = reinterpret_cast<float (*)[TAIL_P]>(acc) + sub * TAIL_G;
        slab_block<8>((-1 - wg) * (RT / 256) + sub, lt, sa, fu.a, step_now, part, [] { __syncthreads(); });
        return;
    }
    // locate (level, bucket, slice) of this work unit
    int l = 0;
    while (l + 1 < meta.num_levels && wg >= bm.wgstart[l + 1]) ++l;
    const int Smax = bm.slices[l];
    const int local = wg - bm.wgstart[l];
    int b = local / Smax;
    const int s = local - b * Smax;
    const int nb = bm.nb[l];
    // Workgroups are dealt round-robin over the 8 XCDs (observed; used for speed only): on an un-sliced level whose
    // bucket count is a multiple of 8 the workgroups of one XCD take CONTIGUOUS buckets.  The segments of neighbouring
    // buckets are neighbours inside every chunk and share 128-byte lines at their seams: read by workgroups of one XCD at
    // about the same time, those lines come from that XCD's L2 the second time instead of twice through the fabric.
    if (LNERF_REDUCE_XCD && Smax == 1 && (nb & 7) == 0) b = (local & 7) * (nb >> 3) + (local >> 3);
    const int tid = threadIdx.x, lane = tid & 63;
    // (uniform, and known to be: everything derived from it -- the wave's items, their chunk addresses -- stays in
    // scalar registers; as a function of threadIdx it was per-lane 64-bit address arithmetic and spilled)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int I = *items_dev;                       // items of the pass 1 that filled the workspace
    I = I < bm.n_items ? I : bm.n_items;
    const uint32_t *tab = segtab + (int64_t)bm.bstart[l] * bm.n_items + b;   // entry of item t: tab[t * nb]
    int S = 1, i0 = 0, i1 = I;
    if (Smax > 1) {  // (uniform per level) a level whose buckets MAY be sliced: count the bucket's records first
        int cnt = 0;
        for (int t = tid; t < I; t += RT) cnt += (int)(tab[(int64_t)t * nb] >> 16);
        cnt = wave_inclusive_sum_i(cnt);
        if (lane == 63) s_red[wave] = cnt;
        __syncthreads();
        int n = 0;
#pragma unroll
        for (int k = 0; k < NW; ++k) n += s_red[k];
        S = active_slices(n, Smax);
        if (s == 0 && tid == 0) bucket_n[bm.bstart[l] + b] = n;   // for the finishing pass
        if (s >= S) return;    // uniform per workgroup
        i0 = (int)(((long long)I * s) / S);
        i1 = (int)(((long long)I * (s + 1)) / S);
    }
    const bool direct = S == 1;        // this workgroup sums the whole bucket: it finishes the rows itself
    const bool fuse = FUSE;            // (whoever finishes a bucket -- its only workgroup, or the last slice to arrive)
    // uniform.  (An active slice always has items: S > 1 means more than REDUCE_SLICE_RECS >= ITEM_RECS records, i.e.
    // at least S items.)
    const bool have = i1 > i0 || !direct;
    if (!have && !fuse) return;  // (a fused bucket without records still owes its rows the Adam step, g = 0)
    constexpr int FB = FixBits<REC>::kBits;
    // from the bound of |value| of the LEVEL (found by pass 1).  One device-scope atomic load: the last workgroup of a
    // closing launch to arrive ZEROES the maxima (tail_arrive) -- ordered behind this read by the arrival, but a plain
    // load the compiler might re-issue later would be a data race on paper
    const FixScale fs = fix_scale(__hip_atomic_load(&gmax[l * CUR_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                                  REC::kPacked ? FB : bm.fix_bits);
    RED_STAMP_INIT();
    const int hsize = meta.offsets[l + 1] - meta.offsets[l];
    const int row0 = b << BK_SHIFT;
    int rows = hsize - row0;
    rows = rows < BK_ROWS ? rows : BK_ROWS;
    const int64_t R0 = (int64_t)meta.offsets[l] + row0;
    // the usual fused case (full bucket, even first row): two rows per lane and access (16 B).  Their parameters and
    // moments are requested BEHIND the record loop and IN FRONT of the barrier that ends it: a wave that is done with its
    // records waits for the slowest wave anyway, and the loads travel meanwhile.  (Requested ahead of the record stream
    // they were measured 35 us slower: the records queue behind them.)
    constexpr int NQ = (BK_ROWS / 2 + RT - 1) / RT;  // row pairs per lane
#ifdef LNERF_EXP_RED_NOADAM     // timing-only experiment build: the workgroup ends behind its record loop
    const bool fast = false;
#else
    const bool fast = fuse && direct && !fu.grad_out && ((R0 | rows) & 1) == 0 && rows == BK_ROWS && (BK_ROWS / 2) % RT == 0;
#endif
    float4 P[NQ], Mv[NQ], V[NQ];
    float4 *p4 = reinterpret_cast<float4 *>(reinterpret_cast<float2 *>(fu.p) + R0);
    float4 *m4 = reinterpret_cast<float4 *>(reinterpret_cast<float2 *>(fu.m) + R0);
    float4 *v4 = reinterpret_cast<float4 *>(reinterpret_cast<float2 *>(fu.v) + R0);
    if (have) {
        const int first = i0 + wave;                                  // this wave's items: first, first + NW, ...
        const int nmy = first < i1 ? (i1 - first + NW - 1) / NW : 0;
        // (the first 64 segment entries are requested before the accumulators are cleared: one round trip hidden)
        uint32_t e_first = 0u;
        if (lane < nmy) e_first = tab[(int64_t)(first + NW * lane) * nb];
        for (int i = tid; i < BK_ROWS * 2; i += RT) acc[i] = 0ll;
        __syncthreads();
        step_now = __builtin_amdgcn_readfirstlane(step_now);   // (returned by now: into a scalar register for the loop)
        RED_STAMP(10);
        const REC *lrec = recs + (int64_t)l * bm.n_items * ITEM_RECS;
        unsigned long long *ua = reinterpret_cast<unsigned long long *>(acc);
        auto add = [&](const REC &r) {
            const uint32_t a0 = r.row_in_bucket();
            atomicAdd(&ua[a0], (unsigned long long)to_fixed<FB>((r.a() * fs.sc_a) * fs.sc_b));
            atomicAdd(&ua[a0 + BK_ROWS], (unsigned long long)to_fixed<FB>((r.b() * fs.sc_a) * fs.sc_b));
        };
        // rounds of loads in flight per lane.  A 12-byte record is three registers: five rounds in flight are what the
        // 64 registers of two resident workgroups leave room for (eight spilled 20-36 bytes per lane to scratch, whose
        // traffic shares the vector-memory queue with the very loads the loop waits for)
        constexpr int U = REC::kPacked ? LNERF_REDUCE_ROUNDS : (LNERF_REDUCE_ROUNDS < 5 ? LNERF_REDUCE_ROUNDS : 5);
#ifdef LNERF_EXP_RED_NOREC   // timing-only experiment build: no record loop
        for (int kb = 0; kb < 0; kb += 64) {
#else
        for (int kb = 0; kb < nmy; kb += 64) {                        // (one pass for up to 64 x 16 = 1024 items)
#endif
            uint32_t e = e_first;
            if (kb > 0) e = kb + lane < nmy ? tab[(int64_t)(first + NW * (kb + lane)) * nb] : 0u;
            const int cnt = nmy - kb < 64 ? nmy - kb : 64;            // segments held by the lanes (uniform)
            const int T = __builtin_amdgcn_readlane(wave_inclusive_sum_i((int)(e >> 16)), 63);   // their records
            // Walk the CONCATENATION of the segments 64 records per round: lane i of a round takes flat record f0 + i, so
            // every lane carries a record whatever the segment sizes are (~32 on a hashed level, thousands on a
            // one-bucket level).  Which segment a lane is in comes from a SCALAR walk: (sj, sp) = first segment that
            // reaches into the round and its flat start; a round visits the 2-3 segments it spans, each visit two scalar
            // readlanes and three vector instructions -- no cross-lane traffic on the LDS pipe, which the two 64-bit
            // atomics of every record need (a binary search through ds_bpermute was 12 us slower).
#if LNERF_REDUCE_WALK == 0
            int sj = 0, sp = 0;
            // record index (inside the level's region) of flat record fb + lane; called with increasing fb
            auto locate = [&](int fb) __attribute__((always_inline)) -> uint32_t {
                const int f = fb + lane;
                uint32_t at = 0u;
                for (;;) {
                    const uint32_t ej = (uint32_t)__builtin_amdgcn_readlane((int)e, sj);
                    const int cj = (int)(ej >> 16);
                    // (chunk of item `first + NW (kb + sj)`, its segment's first slot, minus the flat start)
                    const uint32_t base = (uint32_t)(first + NW * (kb + sj)) * (uint32_t)ITEM_RECS + (ej & 0xFFFFu) -
                                          (uint32_t)sp;
                    at = (f >= sp && f < sp + cj) ? base + (uint32_t)f : at;
                    if (sp + cj >= fb + 64 || sj + 1 >= cnt) break;   // the round ends inside this segment
                    sp += cj;
                    ++sj;
                }
                return f < T ? at : 0u;                               // (slot 0 exists: the load is unconditional)
            };
            // software pipeline over the rounds: U loads are in flight at ALL times -- a round's record is consumed and
            // its register immediately re-armed with the load of the round U ahead (a plain "issue U, consume U" loop
            // drains to zero loads in flight at the end of every batch)
            const int nr = (T + 63) >> 6;                             // rounds (uniform)
            REC r[U];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (u < nr) r[u] = ld_rec(lrec + locate(64 * u));
            for (int rb = 0; rb < nr; rb += U) {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int rd = rb + u;                            // uniform
                    if (rd < nr) {
                        pin_record(r[u]);                             // (keeps the load outside the predicated block)
                        const REC cur = r[u];
                        if (rd + U < nr) r[u] = ld_rec(lrec + locate(64 * (rd + U)));
                        if (64 * rd + lane < T) add(cur);
                    }
                }
            }
#else
            // Which segment a lane's record is in comes from a BITMAP of the segment starts: bit p = "flat record p is the
            // first of its segment"; lane k keeps bits [64 k, 64 k + 64).  A round reads its 64 bits with two scalar
            // readlanes; a lane's segment is the number of starts at or below its position (mbcnt), its record's slot
            // one LDS read of that segment's (chunk slot - flat start) plus its position: ~10 instructions per round
            // instead of a scalar walk over the 2-3 segments a round spans (~150: the pass was bound by instruction issue,
            // not by HBM -- without the Adam phase it took 87 us for 240 MB).  The bitmap covers SUB = 4096 records (64
            // rounds): a window of 64 segments is taken in sub-batches of whole segments with at most SUB records.
            (void)cnt;
            constexpr int SUB = 4096;
            static_assert(ITEM_RECS <= SUB, "a segment must fit a sub-batch");
            const int c = (int)(e >> 16);                             // records of the lane's segment
            const int inc = wave_inclusive_sum_i(c);
            int a = 0, a_base = 0;                                    // first lane / flat start of the sub-batch (uniform)
            while (a_base < T) {
                const bool in = lane >= a && inc - a_base <= SUB;     // (inc is monotone: a contiguous run from lane a)
                const int bnd = a + (int)__popcll(__ballot(in));      // one past the sub-batch's last lane, > a
                const int Ts = __builtin_amdgcn_readlane(inc, bnd - 1) - a_base;   // its records
                const bool seg = in && c > 0;
                const int ci = (int)mbcnt(__ballot(seg));             // index among the non-empty segments
                const int start = inc - c - a_base;                   // flat start inside the sub-batch
                s_bmp[wave][2 * lane] = 0u;
                s_bmp[wave][2 * lane + 1] = 0u;
                if (seg) {
                    atomicOr(&s_bmp[wave][start >> 5], 1u << (start & 31));
                    s_soff[wave][ci] = (uint32_t)(first + NW * (kb + lane)) * (uint32_t)ITEM_RECS + (e & 0xFFFFu) -
                                       (uint32_t)start;
                }
                // the lanes exchange data through LDS: a wave's LDS operations execute in order, but the COMPILER reasons
                // per thread -- without the fence pair a lane that set no bit "knows" its words are still zero and never
                // reads them back (measured: the read was sunk into the `if (seg)` block above)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const int bm_lo = (int)s_bmp[wave][2 * lane], bm_hi = (int)s_bmp[wave][2 * lane + 1];
                int nstart = 0;                                       // segment starts before the round (uniform)
                // record index (inside the level's region) of flat record fb + lane; called with increasing fb
                auto locate = [&](int fb) __attribute__((always_inline)) -> uint32_t {
                    const int k = __builtin_amdgcn_readfirstlane(fb >> 6);
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane(bm_lo, k);
                    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane(bm_hi, k);
                    const unsigned long long m1 = (((unsigned long long)hi << 32) | lo) >> 1;
                    // starts at positions 1..lane = bits below `lane` of (M >> 1); position 0 = bit 0 of M
                    const int below = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32),
                                                                     __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
                    const int j = nstart + (int)(lo & 1u) - 1 + below;
                    nstart += __popc(lo) + __popc(hi);
                    const int f = fb + lane;
                    const uint32_t at = s_soff[wave][j < 0 ? 0 : j] + (uint32_t)f;
                    return f < Ts ? at : 0u;                          // (slot 0 exists: the load is unconditional)
                };
                // software pipeline over the rounds: U loads are in flight at ALL times -- a round's record is consumed
                // and its register immediately re-armed with the load of the round U ahead
                const int nr = (Ts + 63) >> 6;                        // rounds (uniform)
                REC r[U];
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (u < nr) r[u] = ld_rec(lrec + locate(64 * u));
                for (int rb = 0; rb < nr; rb += U) {
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int rd = rb + u;                        // uniform
                        if (rd < nr) {
                            pin_record(r[u]);                         // (keeps the load outside the predicated block)
                            const REC cur = r[u];
                            if (rd + U < nr) r[u] = ld_rec(lrec + locate(64 * (rd + U)));
                            if (64 * rd + lane < Ts) add(cur);
                        }
                    }
                }
                a = bnd;
                a_base += Ts;
            }
#endif
        }
        RED_STAMP(11);
        if (fast) {
#pragma unroll
            for (int j = 0; j < NQ; ++j) {  // all of the lane's loads: six 16-byte loads in flight behind the barrier
                const int q = tid + j * RT;
                P[j] = ld_f4(p4 + q); Mv[j] = ld_f4(m4 + q); V[j] = ld_f4(v4 + q);
            }
        }
        __syncthreads();
        RED_STAMP(12);
    } else if (fast) {
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
            const int q = tid + j * RT;
            P[j] = ld_f4(p4 + q); Mv[j] = ld_f4(m4 + q); V[j] = ld_f4(v4 + q);
        }
    }
    if (!direct) {
        // Sliced bucket: every slice publishes its EXACT 64-bit partial sums as a tile of `partials`; the slice that
        // arrives LAST adds the other tiles to its own sums and finishes the rows like the only workgroup of an unsliced
        // bucket (integer sums: neither the slicing nor the arrival order changes a bit of the result).  The heavily
        // loaded coarse buckets come first in the grid, so this happens early in the launch, under the other buckets'
        // work -- as its own pass behind the launch it was a chain of dependent round trips (~9 us) at the end of the
        // step.  Tiles travel with device-scope (write-through / cache-bypassing) accesses: slices run on different
        // XCDs, whose L2s are not coherent for plain stores, and the addresses are the same every step.
        //
        // ORDERING -- by construction on the ISA, not by C++ memory orders (every atomic below is RELAXED):
        //   writer   tile stores = `global_store_dwordx2 ... sc1` (write-through to device scope); `s_waitcnt vmcnt(0)`:
        //            every store of the wave ACKNOWLEDGED, i.e. visible at device scope; workgroup barrier: true of all
        //            16 waves; then ONE returning `global_atomic_add ... sc0` on the bucket's arrival word.
        //   reader   (the arrival that returned S - 1) its value reaches the other waves through LDS + a barrier, so every
        //            tile load is issued behind the atomic's return; tile loads = `global_load_dwordx2 ... sc1`: they
        //            miss this XCD's non-coherent L2 and see the acknowledged stores.
        // A release / acquire pair at agent scope would be correct by the letter and costs a `buffer_wbl2` -- a write-back
        // of the XCD's whole L2 -- per workgroup: 106 -> 273 us for this pass.  tests/test_abi_cpu.py
        // (test_cross_workgroup_handoffs_are_scoped_accesses) checks the compiled kernel for exactly these instructions
        // and for the absence of L2 write-backs / invalidates, so a compiler that chose otherwise fails the CPU suite.
        unsigned long long *tiles = reinterpret_cast<unsigned long long *>(partials) +
                                    ((int64_t)bm.pstart[l] + (int64_t)b * Smax) * (BK_ROWS * 2);
        unsigned long long *pt = tiles + (int64_t)s * (BK_ROWS * 2);
        const unsigned long long *ul = reinterpret_cast<const unsigned long long *>(acc);
        for (int i = tid; i < BK_ROWS * 2; i += RT)
            __hip_atomic_store(&pt[i], ul[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's part of the tile has been written
        __syncthreads();
        int32_t *arr = slice_arrive + bm.bstart[l] + b;
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(arr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == S - 1) __hip_atomic_store(arr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // clean for the next call
            s_red[0] = old;
        }
        __syncthreads();
        if (s_red[0] != S - 1) return;     // uniform: an earlier arrival, somebody else finishes the bucket
        // (one tile at a time, ALL of the lane's elements of it in flight: element by element the sum was a chain of
        // 8 (S - 1) dependent round trips -- 94 us for a six-slice bucket, the longest workgroup of the launch)
        constexpr int NI = BK_ROWS * 2 / RT, NB = NI < 4 ? NI : 4;   // (four 64-bit loads in flight: no spill at 64 registers)
        for (int k0 = 0; k0 < NI; k0 += NB) {
            unsigned long long q[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) q[k] = ul[tid + (k0 + k) * RT];
            for (int s2 = 0; s2 < S; ++s2) {
                if (s2 == s) continue;     // uniform
                const unsigned long long *ot = tiles + (int64_t)s2 * (BK_ROWS * 2) + tid + k0 * RT;
                unsigned long long t[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k)
                    t[k] = __hip_atomic_load(&ot[k * RT], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int k = 0; k < NB; ++k) q[k] += t[k];
            }
#pragma unroll
            for (int k = 0; k < NB; ++k) acc[tid + (k0 + k) * RT] = (long long)q[k];
        }
        __syncthreads();
    }
    float *dst = dtable + R0 * 2;
#ifdef LNERF_EXP_RED_NOADAM
    if (fuse) return;
#endif
    if (fuse) {
        AdamArgs a = fu.a;
        if (have_step) adam_bias_at(a, __builtin_amdgcn_readfirstlane(step_now));   // (the closing launch: see the kernel)
        else adam_bias(a);
        a.zero_grad = 0;
        float2 *p2 = reinterpret_cast<float2 *>(fu.p) + R0, *m2 = reinterpret_cast<float2 *>(fu.m) + R0;
        float2 *v2 = reinterpret_cast<float2 *>(fu.v) + R0;
        uint32_t *sh = fu.shadow ? reinterpret_cast<uint32_t *>(fu.shadow) + R0 : nullptr;
        auto grad_of = [&](int r, float &g0, float &g1) {
            g0 = 0.f; g1 = 0.f;
            if (have) {
                g0 = ((float)acc[r] * fs.un_a) * fs.un_b;
                g1 = ((float)acc[r + BK_ROWS] * fs.un_a) * fs.un_b;
            }
        };
        if (fu.grad_out) {  // gradient output in the wire format: one bf16 pair per row
            uint32_t *go = reinterpret_cast<uint32_t *>(fu.grad_out) + R0;
            for (int r = tid; r < rows; r += RT) {
                float g0, g1;
                grad_of(r, g0, g1);
                go[r] = (uint32_t)f32_to_bf16(g0) | ((uint32_t)f32_to_bf16(g1) << 16);
            }
            return;
        }
        if (fast) {
            uint2 *sh2 = reinterpret_cast<uint2 *>(sh);
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int q = tid + j * RT;
                float ga, gb, gc, gd;
                grad_of(2 * q, ga, gb);
                grad_of(2 * q + 1, gc, gd);
                adam_one(P[j].x, ga, Mv[j].x, V[j].x, a);
                adam_one(P[j].y, gb, Mv[j].y, V[j].y, a);
                adam_one(P[j].z, gc, Mv[j].z, V[j].z, a);
                adam_one(P[j].w, gd, Mv[j].w, V[j].w, a);
                st_f4(p4 + q, P[j]); st_f4(m4 + q, Mv[j]); st_f4(v4 + q, V[j]);
                if (sh) {
                    uint2 w;
                    w.x = (uint32_t)f32_to_bf16(P[j].x) | ((uint32_t)f32_to_bf16(P[j].y) << 16);
                    w.y = (uint32_t)f32_to_bf16(P[j].z) | ((uint32_t)f32_to_bf16(P[j].w) << 16);
                    sh2[q] = w;
                }
            }
            RED_STAMP(13);
            RED_STAMP_FLUSH();
            return;
        }
        for (int r = tid; r < rows; r += RT) {
            float2 Pr = p2[r], Mr = m2[r], Vr = v2[r];
            float g0, g1;
            grad_of(r, g0, g1);
            adam_one(Pr.x, g0, Mr.x, Vr.x, a);
            adam_one(Pr.y, g1, Mr.y, Vr.y, a);
            p2[r] = Pr; m2[r] = Mr; v2[r] = Vr;
            if (sh) sh[r] = (uint32_t)f32_to_bf16(Pr.x) | ((uint32_t)f32_to_bf16(Pr.y) << 16);
        }
        return;
    }
    // sole owner of these rows in this launch: plain read-modify-write, 8 B per lane
    for (int r = tid; r < rows; r += RT) {
        float2 d = reinterpret_cast<float2 *>(dst)[r];
        d.x += ((float)acc[r] * fs.un_a) * fs.un_b;
        d.y += ((float)acc[r + BK_ROWS] * fs.un_a) * fs.un_b;
        reinterpret_cast<float2 *>(dst)[r] = d;
    }
}

// One workgroup per (bucket, slice) unit.  (PERSISTENT workgroups striding over the units were built and measured: the
// loop keeps the three kernel-argument structs live across iterations, 77 VGPRs spill at the 64 the two-workgroups-per-CU
// occupancy allows, and the pass went from 0.211 to 0.27 ms per scatter call: profiles/r03_exp_scatter.jsonl.)
#ifndef LNERF_FUSED_RT          // threads per workgroup of the fused pass (experiment knob: 512 with LNERF_BK_SHIFT = 11)
#define LNERF_FUSED_RT 1024
#endif
template <int RT, typename REC, bool FUSE>
__global__ void __launch_bounds__(RT, (LNERF_BK_SHIFT < 12 && RT == 512) ? 8 : RT / 128)
k_scatter_reduce(GridMeta meta, BucketMeta bm, const int32_t *__restrict__ items_dev, const uint32_t *__restrict__ segtab,
                 int32_t *__restrict__ bucket_n, int32_t *__restrict__ slice_arrive, unsigned int *gmax,
                 const REC *__restrict__ recs, float *__restrict__ dtable, long long *__restrict__ partials, int wg_lo,
                 FusedUpdate fu, TailJob tj) {
    // (closing the step: the counter is read ONCE per wave, with a device-scope atomic load, before anything else --
    // the last workgroup of the launch to arrive rewrites it)
#ifdef LNERF_STAMPS
    const unsigned long long wg_t0 = wall_clock64();
#endif
    int32_t step_now = 0;
    const bool closing = FUSE && (tj.do_tick || tj.clear_gmax || tj.blocks > 0);
    // (requested here, consumed behind the workgroup's first barrier: no stall in front of the record stream)
    if (closing && fu.a.step_dev) step_now = __hip_atomic_load(fu.a.step_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int unit = (int)blockIdx.x - (FUSE ? tj.blocks : 0);
    // heaviest first: the un-merged fine levels carry the most records per bucket; taken in level order they ran LAST
    // and the launch's tail was its heaviest workgroups (the sliced coarse levels keep their place at the front)
    if (unit + wg_lo >= tj.rev_lo && unit + wg_lo < tj.rev_hi) unit = tj.rev_lo + (tj.rev_hi - 1 - (unit + wg_lo)) - wg_lo;
    scatter_reduce_one<RT, REC, FUSE>(unit < 0 ? unit : unit + wg_lo, meta, bm, items_dev, segtab, bucket_n, slice_arrive,
                                      gmax, recs, dtable, partials, fu, tj.sa, step_now, closing && fu.a.step_dev != nullptr);
#ifdef LNERF_STAMPS
    if (blockIdx.x < 4096) {   // (exit = the LAST wave's, with its stores acknowledged: the slot is free after that)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if ((threadIdx.x & 63) == 0) atomicMax(&g_wg_log[4 * blockIdx.x + 1], (unsigned long long)wall_clock64());
        if (threadIdx.x == 0) {
            unsigned int hw, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            g_wg_log[4 * blockIdx.x] = wg_t0;
            g_wg_log[4 * blockIdx.x + 2] = ((unsigned long long)xcc << 32) | hw;
            g_wg_log[4 * blockIdx.x + 3] = (unsigned long long)(unsigned int)unit;
        }
    }
#endif
    if (closing && (tj.do_tick || tj.clear_gmax) && threadIdx.x == 0)
        // (this wave is done.  The workgroup's other waves requested the counter as their first instruction and the
        // level maximum in front of the record loop; a workgroup's barriers wait for a wave's outstanding loads, and
        // a workgroup that leaves before its first barrier has not used either value)
        tail_arrive(tj.arrive, (int)gridDim.x, (int)blockIdx.x, tj.tick, step_now, gmax, tj.do_tick, tj.clear_gmax);
}

// the slab blocks + the closing arrival as a launch of their own (lnerf_step_tail)
__global__ void __launch_bounds__(256)
k_step_tail(unsigned int *__restrict__ gmax, AdamArgs a, SlabAdam sa, int32_t *__restrict__ tick,
            int32_t *__restrict__ arrive, int do_tick, int clear_gmax) {
    int32_t step_now = 0;
    if (a.step_dev) step_now = __hip_atomic_load(a.step_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __shared__ float part[TAIL_G][TAIL_P];
    if (sa.slabs) slab_block<TAIL_SLABS_PER_LANE>((int)blockIdx.x, (int)threadIdx.x, sa, a, step_now, part, [] { __syncthreads(); });
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (do_tick || clear_gmax) {
        __syncthreads();
        if (threadIdx.x == 0) tail_arrive(arrive, (int)gridDim.x, (int)blockIdx.x, tick, step_now, gmax, do_tick, clear_gmax);
    }
}

extern int g_mlp_fwd_blocks, g_mlp_fwd_wps, g_mlp_bwd_blocks, g_mlp_bwd_variant;  // mlp.hip

// levels up to this resolution merge per-wave runs before binning (tunable: lnerf_set_tuning)
static int g_compact_max_res = 512;
// gather: fetch x-adjacent vertices with one load where they are adjacent rows (2: also aligned groups of four rows)
static int g_gather_pairs = 2;
// gather variant 2: workgroups per XCD (each strides over the tiles of its XCD's levels)
static int g_gather_wgs_per_xcd = 256;
// gather: levels with resolution <= this fetch a cell's vertices once per run of lanes in that cell (0 = off)
static int g_gather_dedup_res = 512;
// persistent workgroups of the binning pass per CU (3 fit its 44 KiB of LDS with the 8-byte records)
static int g_bin_per_cu = 3;
// persistent workgroups of the binning pass (0 = 256 CUs x g_bin_per_cu); rounded down to a multiple of the level count
static int g_bin_wgs = 0;
// drop contributions that are exactly zero (samples behind a ray's termination point)
static int g_skip_zero = 1;
// threads per workgroup of the reduce pass (512 or 1024; two 64 KiB workgroups fit a CU either way)
static int g_reduce_threads = 1024;
// level groups of the whole-frame scatter: bin(group) -> reduce(group) per group (1 = bin everything, then reduce)
static int g_scatter_groups = 1;

// workspace: [header: level maxima | item count | record count per bucket] [segment table] [record chunks] [partial tiles]
static_assert(BK_MAX_PER_LEVEL == 256 && HDR_BUCKETN_OFF <= LNERF_SCATTER_ZERO_HEAD_BYTES,
              "the counters of the header must lie inside the head a caller zeroes");
static size_t header_bytes(int n_buckets) {
    return (HDR_BUCKETN_OFF + (size_t)n_buckets * sizeof(int32_t) + 4095) / 4096 * 4096;
}

struct ScatterPlan {
    int buckets, wgs;   // buckets / pass-2 workgroups over all levels
    int ptiles;         // partial-sum tiles (sliced levels: buckets x slices)
    int fbuckets;       // buckets of sliced levels (grid of the finishing pass)
    size_t header_bytes, seg_bytes, rec_bytes, partial_bytes;
    size_t total() const { return header_bytes + seg_bytes + rec_bytes + partial_bytes; }
};

static int fill_bucket_meta(const GridMeta &meta, int64_t m_host, BucketMeta &bm, ScatterPlan &plan) {
    int total_buckets = 0, total_wgs = 0, total_ptiles = 0, total_fb = 0;
    const int64_t n_items = m_host > 0 ? div_up(m_host, (int64_t)ITEM_SAMPLES) : 1;
    if (n_items >= (1 << 19)) return -1;   // (pass 2 addresses a level's records with 32-bit record indices)
    for (int l = 0; l < meta.num_levels; ++l) {
        const int64_t hsize = meta.offsets[l + 1] - meta.offsets[l];
        const int nb = (int)div_up(hsize, BK_ROWS);
        if (nb > BK_MAX_PER_LEVEL) return -1;
        const int64_t per_bucket = div_up(8 * m_host, nb);   // worst case under a uniform spread
        int slices = (int)((per_bucket + 65535) / 65536);    // <= ~64 Ki records per pass-2 workgroup
        if (slices < 1) slices = 1;
        if (slices > 64) slices = 64;
        bm.nb[l] = nb;
        bm.bstart[l] = total_buckets;
        bm.slices[l] = slices;
        bm.compact[l] = meta.res[l] <= g_compact_max_res ? 1 : 0;
        bm.wgstart[l] = total_wgs;
        bm.pstart[l] = slices > 1 ? total_ptiles : -1;
        bm.fstart[l] = slices > 1 ? total_fb : -1;
        if (slices > 1) {
            total_ptiles += nb * slices;
            total_fb += nb;
        }
        total_buckets += nb;
        total_wgs += nb * slices;
    }
    bm.bstart[meta.num_levels] = total_buckets;
    bm.wgstart[meta.num_levels] = total_wgs;
    bm.n_items = (int)n_items;
    // 12-byte records, exact sums: the addends of a bucket sum to at most m_host x the level's bound (the weights of a
    // sample's 8 vertices sum to 1; a merged run's bound is 64 x the largest |g| and it stands for up to 64 samples), so
    // bits + ceil(log2(m_host)) <= 62 keeps every int64 sum exact whatever the input: 44 bits up to 2^18 samples, 42 at
    // the bench's 640 Ki, 39 with eight views in a batch (the 8-byte records use 30 bits: exact below 2^32 samples)
    int lg = 1;
    while (((int64_t)1 << lg) < m_host) ++lg;
    bm.fix_bits = 62 - lg < 44 ? 62 - lg : 44;
    plan.buckets = total_buckets;
    plan.wgs = total_wgs;
    plan.ptiles = total_ptiles;
    plan.fbuckets = total_fb;
    plan.header_bytes = header_bytes(total_buckets);
    plan.seg_bytes = ((size_t)total_buckets * (size_t)n_items * sizeof(uint32_t) + 4095) / 4096 * 4096;
    // one chunk of ITEM_RECS slots per (level, item): exactly what the item's samples can emit (sized for the 12-byte
    // records; the packed ones use two thirds of it)
    plan.rec_bytes = (size_t)meta.num_levels * (size_t)n_items * ITEM_RECS * sizeof(Rec12);
    plan.partial_bytes = (size_t)total_ptiles * BK_ROWS * 2 * sizeof(long long);
    return 0;
}

static int fill_meta(const char *who, GridMeta &meta, int num_levels, int level_dim, const int32_t *offsets_host,
                     const float *scales_host, const int32_t *res_host, int layout_flags = 0) {
    LNERF_REQUIRE(num_levels >= 1 && num_levels <= LNERF_MAX_LEVELS, "%s: num_levels out of range (%d)", who,
                  num_levels);
    LNERF_REQUIRE(level_dim == 2, "%s: only level_dim == 2 is built (got %d)", who, level_dim);
    LNERF_REQUIRE(offsets_host && scales_host && res_host, "%s: null level metadata", who);
    meta.num_levels = num_levels;
    LNERF_REQUIRE((layout_flags & (LNERF_GRID_BLOCKED | LNERF_GRID_TILED)) != (LNERF_GRID_BLOCKED | LNERF_GRID_TILED),
                  "%s: LNERF_GRID_BLOCKED and LNERF_GRID_TILED exclude each other", who);
    const int blocked = (layout_flags & LNERF_GRID_BLOCKED) ? 1 : 0;
    meta.blocked = blocked ? 1 : ((layout_flags & LNERF_GRID_TILED) ? 2 : 0);   // layout of the levels beyond their table: 0 hash, 1 blocked, 2 tiled
    for (int l = 0; l <= num_levels; ++l) meta.offsets[l] = offsets_host[l];
    for (int l = 0; l < num_levels; ++l) {
        LNERF_REQUIRE(offsets_host[l + 1] > offsets_host[l], "%s: empty level %d", who, l);
        LNERF_REQUIRE(!blocked || offsets_host[l + 1] - offsets_host[l] >= 16, "%s: blocked layout needs >= 16 rows per level", who);
        LNERF_REQUIRE(res_host[l] >= 1 && res_host[l] <= 1 << 20, "%s: bad resolution at level %d", who, l);
        meta.scales[l] = scales_host[l];
        meta.res[l] = res_host[l];
    }
    return LNERF_OK;
}

static void launch_dims(int variant, int L, int64_t m_host, dim3 &grid) {
    const int64_t tiles = div_up(m_host, 256);
    if (variant == 0) {
        int64_t gx = tiles < 1 ? 1 : tiles;
        if (gx > 2048) gx = 2048;
        grid = dim3((unsigned)gx, (unsigned)L, 1);
    } else {
        const int lv_per_xcd = (L + 7) / 8;
        int64_t per_level = tiles < 1 ? 1 : tiles;
        if (per_level > 256) per_level = 256;  // workgroups per level
        grid = dim3((unsigned)(8 * lv_per_xcd * per_level), 1, 1);
    }
}

}  // namespace lnerf

using namespace lnerf;

extern "C" {

int lnerf_grid_encode_forward(const float *xyzs, float bound, const void *table, int table_dtype, int num_levels,
                              int level_dim, const int32_t *offsets_host, const float *scales_host,
                              const int32_t *res_host, int64_t m_host, const int32_t *m_dev, int64_t level_stride,
                              void *feat, int feat_dtype, int variant, lnerf_stream_t stream) {
    GridMeta meta;
    const int blocked = variant & (LNERF_GRID_BLOCKED | LNERF_GRID_TILED);
    variant &= ~(LNERF_GRID_BLOCKED | LNERF_GRID_TILED);
    int rc = fill_meta("grid_encode_forward", meta, num_levels, level_dim, offsets_host, scales_host, res_host, blocked);
    if (rc) return rc;
    LNERF_REQUIRE(m_host >= 0 && level_stride >= m_host, "grid_encode_forward: need 0 <= m_host <= level_stride");
    LNERF_REQUIRE(bound > 0.f, "grid_encode_forward: bound must be > 0");
    LNERF_REQUIRE(variant >= 0 && variant <= 2, "grid_encode_forward: unknown variant %d", variant);
#ifndef LNERF_EXPERIMENTS   // (XCD-pinned levels / XCD-owned level sets: measured no faster; experiment builds only)
    LNERF_REQUIRE(variant == 0, "grid_encode_forward: variant %d is an experiment variant (build with -DLNERF_EXPERIMENTS)", variant);
#endif
    LNERF_REQUIRE((table_dtype == LNERF_F32 || table_dtype == LNERF_BF16) &&
                      (feat_dtype == LNERF_F32 || feat_dtype == LNERF_BF16),
                  "grid_encode_forward: bad dtype tag");
    if (m_host == 0) return LNERF_OK;
    LNERF_REQUIRE(xyzs && table && feat, "grid_encode_forward: null pointer");
    dim3 grid;
    launch_dims(variant == 2 ? 0 : variant, num_levels, m_host, grid);
    XcdPlan plan;
    memset(&plan, 0, sizeof(plan));
    if (variant == 2) {
        // longest-processing-time assignment of levels to XCDs; cost ~ cache lines a sample touches on the level
        double cost[LNERF_MAX_LEVELS], load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        int order[LNERF_MAX_LEVELS];
        for (int l = 0; l < num_levels; ++l) {
            const double r = (double)res_host[l];
            cost[l] = res_host[l] > g_gather_dedup_res ? 1.0 : (r < 64 ? 0.05 : r / (double)(g_gather_dedup_res > 0 ? g_gather_dedup_res : 512) * 0.9);
            order[l] = l;
        }
        for (int a = 0; a < num_levels; ++a)
            for (int b = a + 1; b < num_levels; ++b)
                if (cost[order[b]] > cost[order[a]]) { const int t = order[a]; order[a] = order[b]; order[b] = t; }
        for (int a = 0; a < num_levels; ++a) {
            int best = -1;
            for (int x = 0; x < 8; ++x)
                if (plan.n[x] < LNERF_MAX_LEVELS / 8 + 2 && (best < 0 || load[x] < load[best])) best = x;
            plan.lv[best][plan.n[best]++] = order[a];
            load[best] += cost[order[a]];
        }
        grid = dim3((unsigned)(8 * g_gather_wgs_per_xcd), 1, 1);
    }
    hipStream_t s = as_stream(stream);
#define LAUNCH_FWD(TT, TO)                                                                                         \
    hipLaunchKernelGGL((k_grid_forward<TT, TO>), grid, dim3(256), 0, s, xyzs, bound, (const TT *)table, meta, m_host, \
                       m_dev, level_stride, (TO *)feat, variant, g_gather_pairs, g_gather_dedup_res, plan)
    if (table_dtype == LNERF_F32 && feat_dtype == LNERF_F32) LAUNCH_FWD(float, float);
    else if (table_dtype == LNERF_F32) LAUNCH_FWD(float, uint16_t);
    else if (feat_dtype == LNERF_F32) LAUNCH_FWD(uint16_t, float);
    else LAUNCH_FWD(uint16_t, uint16_t);
#undef LAUNCH_FWD
    LNERF_CHECK_LAUNCH("grid_encode_forward");
    return LNERF_OK;
}

int lnerf_set_tuning(const char *key, int value) {
    LNERF_REQUIRE(key, "set_tuning: null key");
    if (strcmp(key, "scatter_compact_max_res") == 0) {
        LNERF_REQUIRE(value >= 0, "set_tuning: scatter_compact_max_res must be >= 0");
        g_compact_max_res = value;
        return LNERF_OK;
    }
    if (strcmp(key, "scatter_bin_wgs") == 0) {
        LNERF_REQUIRE(value >= 0 && value <= 65535, "set_tuning: scatter_bin_wgs out of range");
        g_bin_wgs = value;
        return LNERF_OK;
    }
    if (strcmp(key, "scatter_bin_per_cu") == 0) {
        LNERF_REQUIRE(value >= 1 && value <= 4, "set_tuning: scatter_bin_per_cu must be in 1 .. 4");
        g_bin_per_cu = value;
        return LNERF_OK;
    }
    if (strcmp(key, "gather_dedup_max_res") == 0) {
        LNERF_REQUIRE(value >= 0, "set_tuning: gather_dedup_max_res must be >= 0");
        g_gather_dedup_res = value;
        return LNERF_OK;
    }
    if (strcmp(key, "gather_wgs_per_xcd") == 0) {
        LNERF_REQUIRE(value >= 1 && value <= 4096, "set_tuning: gather_wgs_per_xcd out of range");
        g_gather_wgs_per_xcd = value;
        return LNERF_OK;
    }
    if (strcmp(key, "gather_pair_loads") == 0) {
        LNERF_REQUIRE(value >= 0 && value <= 2, "set_tuning: gather_pair_loads must be 0, 1 or 2");
        g_gather_pairs = value;
        return LNERF_OK;
    }
    if (strcmp(key, "mlp_bwd_variant") == 0) {
        LNERF_REQUIRE(value >= 0 && value <= 2, "set_tuning: mlp_bwd_variant must be 0, 1 or 2");
#ifndef LNERF_EXPERIMENTS
        LNERF_REQUIRE(value == 0, "set_tuning: mlp_bwd_variant %d is an experiment variant (build with -DLNERF_EXPERIMENTS)", value);
#endif
        g_mlp_bwd_variant = value;
        return LNERF_OK;
    }
    if (strcmp(key, "mlp_bwd_blocks") == 0) {
        LNERF_REQUIRE(value >= 1 && value <= 512, "set_tuning: mlp_bwd_blocks must be in 1 .. 512");
        g_mlp_bwd_blocks = value;
        return LNERF_OK;
    }
    if (strcmp(key, "mlp_fwd_wps") == 0) {
        LNERF_REQUIRE(value >= 2 && value <= 4, "set_tuning: mlp_fwd_wps must be 2, 3 or 4");
#ifndef LNERF_EXPERIMENTS
        LNERF_REQUIRE(value <= 3, "set_tuning: mlp_fwd_wps 4 is an experiment variant (build with -DLNERF_EXPERIMENTS)");
#endif
        g_mlp_fwd_wps = value;
        return LNERF_OK;
    }
    if (strcmp(key, "mlp_fwd_blocks") == 0) {
        LNERF_REQUIRE(value >= 1 && value <= 65535, "set_tuning: mlp_fwd_blocks out of range");
        g_mlp_fwd_blocks = value;
        return LNERF_OK;
    }
    if (strcmp(key, "scatter_level_groups") == 0) {
        LNERF_REQUIRE(value >= 1 && value <= LNERF_MAX_LEVELS, "set_tuning: scatter_level_groups out of range");
        g_scatter_groups = value;
        return LNERF_OK;
    }
    if (strcmp(key, "scatter_skip_zero") == 0) {
        g_skip_zero = value ? 1 : 0;
        return LNERF_OK;
    }
    if (strcmp(key, "scatter_reduce_threads") == 0) {
        LNERF_REQUIRE(value == 512 || value == 1024, "set_tuning: scatter_reduce_threads must be 512 or 1024");
        g_reduce_threads = value;
        return LNERF_OK;
    }
    set_error("set_tuning: unknown key '%s'", key);
    return LNERF_ERR_INVALID_ARG;
}

#ifdef LNERF_STAMPS
// diagnostic builds only: read (and clear) the per-phase shader-clock totals of the binning pass
int lnerf_debug_wg_log(unsigned long long *out, int words) {   // reads AND clears the log
    static unsigned long long zero[4 * 4096];
    if (words > 4 * 4096) words = 4 * 4096;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_log), (size_t)words * 8) != hipSuccess) return LNERF_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wg_log), zero, sizeof(zero)) != hipSuccess) return LNERF_ERR_HIP;
    return LNERF_OK;
}
int lnerf_debug_bin_stamps(unsigned long long *out16) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bin_stamps), sizeof(z)) != hipSuccess) return LNERF_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bin_stamps), z, sizeof(z)) != hipSuccess) return LNERF_ERR_HIP;
    return LNERF_OK;
}
#endif

size_t lnerf_grid_scatter_clear_bytes(int num_levels, const int32_t *offsets_host, int64_t m_host) {
    if (num_levels < 1 || num_levels > LNERF_MAX_LEVELS || !offsets_host || m_host < 0) return 0;
    GridMeta meta;
    meta.num_levels = num_levels;
    meta.blocked = 0;
    for (int l = 0; l <= num_levels; ++l) meta.offsets[l] = offsets_host[l];
    for (int l = 0; l < num_levels; ++l) meta.res[l] = 0;
    BucketMeta bm;
    ScatterPlan plan;
    if (fill_bucket_meta(meta, m_host, bm, plan) != 0) return 0;
    return HDR_GMAX_BYTES;   // the level maxima (pass 1 raises them with atomicMax)
}

size_t lnerf_grid_encode_backward_workspace_bytes(int num_levels, const int32_t *offsets_host, int64_t m_host) {
    if (num_levels < 1 || num_levels > LNERF_MAX_LEVELS || !offsets_host || m_host < 0) return 0;
    GridMeta meta;
    meta.num_levels = num_levels;
    meta.blocked = 0;
    for (int l = 0; l <= num_levels; ++l) meta.offsets[l] = offsets_host[l];
    for (int l = 0; l < num_levels; ++l) meta.res[l] = 0;
    BucketMeta bm;
    ScatterPlan plan;
    if (fill_bucket_meta(meta, m_host, bm, plan) != 0) return 0;
    return plan.total();  // header, segment table, record chunks, partial-sum tiles of the sliced levels
}

// fu == nullptr: dtable += scatter.  fu != nullptr: every row's Adam step is applied by whichever kernel finishes its
// sum (pass 2 on unsliced levels, the finishing pass on sliced ones); dtable only carries overflow records.
// phases: 1 = pass 1 (binning, all levels; clears the cursors), 2 = pass 2 + finishing pass of levels [lv_lo, lv_hi)
static int scatter_backward(const float *xyzs, float bound, const void *dfeat, int dfeat_dtype, int num_levels,
                            int level_dim, const int32_t *offsets_host, const float *scales_host,
                            const int32_t *res_host, int64_t m_host, const int32_t *m_dev, int64_t level_stride,
                            float *dtable, int variant, void *workspace, size_t workspace_bytes,
                            lnerf_stream_t stream, FusedUpdate *fu, int phases = 3, int lv_lo = 0, int lv_hi = -1,
                            const TailJob *tail = nullptr) {
    // LNERF_SCATTER_CLEARED: the caller zeroed the head of the workspace (lnerf_grid_scatter_clear_bytes()) with
    // something it was launching anyway -- the fill dispatch of this call is skipped
    const bool cleared = (variant & LNERF_SCATTER_CLEARED) != 0;
    // (LNERF_SCATTER_DEFER_FINISH: accepted, without effect -- pass 2 finishes the sliced buckets itself)
    const int blocked = variant & (LNERF_GRID_BLOCKED | LNERF_GRID_TILED);
    variant &= ~(LNERF_SCATTER_CLEARED | LNERF_SCATTER_DEFER_FINISH | LNERF_GRID_BLOCKED | LNERF_GRID_TILED);
    GridMeta meta;
    int rc = fill_meta("grid_encode_backward", meta, num_levels, level_dim, offsets_host, scales_host, res_host, blocked);
    if (rc) return rc;
    LNERF_REQUIRE(m_host >= 0 && level_stride >= m_host, "grid_encode_backward: need 0 <= m_host <= level_stride");
    LNERF_REQUIRE(m_host < (1ll << 30), "grid_encode_backward: m_host must be below 2^30 samples");
    LNERF_REQUIRE(variant < 2 || m_host < (1ll << 28), "grid_encode_backward: the bucketed scatter takes m_host below 2^28");
    LNERF_REQUIRE(bound > 0.f, "grid_encode_backward: bound must be > 0");
    LNERF_REQUIRE(variant >= 0 && variant <= 3, "grid_encode_backward: unknown variant %d", variant);
    LNERF_REQUIRE(dfeat_dtype == LNERF_F32, "grid_encode_backward: dfeat must be f32");
    LNERF_REQUIRE(!fu || (variant >= 2 && m_host > 0), "grid_encode_backward_adam: needs variant 2/3 and m_host > 0");
    if (m_host == 0) return LNERF_OK;
    LNERF_REQUIRE(dtable && (!(phases & 1) || (xyzs && dfeat)), "grid_encode_backward: null pointer");
    if (lv_hi < 0) lv_hi = num_levels;
    LNERF_REQUIRE(lv_lo >= 0 && lv_lo <= lv_hi && lv_hi <= num_levels, "grid_encode_backward: bad level range");
    LNERF_REQUIRE(phases == 3 || variant >= 2, "grid_encode_backward: the split form needs the bucketed scatter");
    hipStream_t s = as_stream(stream);
    dim3 grid;
    if (variant < 2) {
        launch_dims(variant, num_levels, m_host, grid);
        hipLaunchKernelGGL((k_grid_backward_atomic<float>), grid, dim3(256), 0, s, xyzs, bound, (const float *)dfeat,
                           meta, m_host, m_dev, level_stride, dtable, variant);
        LNERF_CHECK_LAUNCH("grid_encode_backward");
        return LNERF_OK;
    }
    BucketMeta bm;
    ScatterPlan plan;
    LNERF_REQUIRE(fill_bucket_meta(meta, m_host, bm, plan) == 0,
                  "grid_encode_backward: level too large for the bucketed scatter (use variant 0/1)");
    const size_t need = plan.total();
    LNERF_REQUIRE(workspace && workspace_bytes >= need, "grid_encode_backward: workspace too small (%zu < %zu)",
                  workspace_bytes, need);
    LNERF_REQUIRE(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)dtable & 15) == 0,
                  "grid_encode_backward: workspace/dtable must be 16-byte aligned");
    char *wsb = (char *)workspace;
    unsigned int *gmax = (unsigned int *)wsb;
    int32_t *items_dev = (int32_t *)(wsb + HDR_ITEMS_OFF);
    int32_t *slice_arrive = (int32_t *)(wsb + HDR_SLICE_ARRIVE_OFF);
    int32_t *bucket_n = (int32_t *)(wsb + HDR_BUCKETN_OFF);
    uint32_t *segtab = (uint32_t *)(wsb + plan.header_bytes);
    void *rec = wsb + plan.header_bytes + plan.seg_bytes;
    long long *partials = (long long *)(wsb + plan.header_bytes + plan.seg_bytes + plan.rec_bytes);
    const bool packed = variant == 3;
    if ((phases & 1) && !cleared && hipMemsetAsync(gmax, 0, HDR_GMAX_BYTES, s) != hipSuccess) {
        set_error("grid_encode_backward: hipMemsetAsync failed");
        return LNERF_ERR_HIP;
    }
    auto launch_bin = [&](int l0, int l1) {
        // persistent: G workgroups, G a multiple of the launch's level count (item k of a workgroup: next tile group, next level)
        const int nl = l1 - l0;
        const int wgs = g_bin_wgs > 0 ? g_bin_wgs : 256 * g_bin_per_cu;
        int64_t G = (int64_t)(wgs / nl) * nl;
        const int64_t items = div_up(m_host, (int64_t)BIN_T) * nl;
        if (G > items) G = items;
        if (G < nl) G = nl;
        const dim3 g((unsigned)G, 1, 1);
        if (packed)
            hipLaunchKernelGGL((k_scatter_bin<Rec8>), g, dim3(BIN_T), 0, s, xyzs, bound, (const float *)dfeat, meta, bm,
                               m_host, m_dev, level_stride, gmax, items_dev, segtab, (Rec8 *)rec, g_skip_zero, l0, l1);
        else
            hipLaunchKernelGGL((k_scatter_bin<Rec12>), g, dim3(BIN_T), 0, s, xyzs, bound, (const float *)dfeat, meta, bm,
                               m_host, m_dev, level_stride, gmax, items_dev, segtab, (Rec12 *)rec, g_skip_zero, l0, l1);
    };
    FusedUpdate fu0;
    memset(&fu0, 0, sizeof(fu0));
    if (fu) fu0 = *fu;
    TailJob tj0;
    memset(&tj0, 0, sizeof(tj0));
    if (tail) {
        LNERF_REQUIRE(fu && !fu->grad_out && phases == 3 && lv_lo == 0 && lv_hi == num_levels && g_scatter_groups <= 1,
                      "grid_encode_backward: the closing form needs the fused whole-table call");
        tj0 = *tail;
        tj0.arrive = (int32_t *)(wsb + HDR_ARRIVE_OFF);
        tj0.blocks = tj0.sa.slabs ? (int)div_up(div_up(MLP_SLAB, TAIL_P), LNERF_FUSED_RT / 256) : 0;   // four slab blocks per workgroup
    }
#ifndef LNERF_REDUCE_ORDER
#define LNERF_REDUCE_ORDER 1
#endif
    if (LNERF_REDUCE_ORDER) {   // whole-table launches only (a level-range launch keeps the plain order)
        int lf = 0;
        for (int l = 0; l < num_levels; ++l) if (bm.slices[l] > 1) lf = l + 1;
        tj0.rev_lo = bm.wgstart[lf];
        tj0.rev_hi = bm.wgstart[num_levels];
    }
    auto launch_reduce = [&](hipStream_t st, int l0, int l1) {
        const int w0 = bm.wgstart[l0], w1 = bm.wgstart[l1];
        if (l0 != 0 || l1 != num_levels) { tj0.rev_lo = tj0.rev_hi = 0; }
        if (w1 <= w0 && tj0.blocks == 0) return;
#define LAUNCH_RED(T, REC, FUSE)                                                                                  \
    hipLaunchKernelGGL((k_scatter_reduce<T, REC, FUSE>), dim3((unsigned)(w1 - w0 + (FUSE ? tj0.blocks : 0))), dim3(T), 0,  \
                       st, meta, bm, items_dev, segtab, bucket_n, slice_arrive, gmax, (const REC *)rec, dtable, partials, \
                       w0, fu0, tj0)
        // (the fused pass with 512-thread workgroups: 118 us against 108, profiles/r03_exp_scatter.jsonl)
        if (fu && packed) LAUNCH_RED(LNERF_FUSED_RT, Rec8, true);
        else if (fu) LAUNCH_RED(LNERF_FUSED_RT, Rec12, true);
        else if (packed && g_reduce_threads == 512) LAUNCH_RED(512, Rec8, false);
        else if (packed) LAUNCH_RED(1024, Rec8, false);
        else if (g_reduce_threads == 512) LAUNCH_RED(512, Rec12, false);
        else LAUNCH_RED(1024, Rec12, false);
#undef LAUNCH_RED
    };
    if (phases == 3 && g_scatter_groups > 1 && lv_lo == 0 && lv_hi == num_levels) {
        // level GROUPS: bin(group) -> reduce(group) -> bin(next group) ...  A group's records (1/groups of the 216 MB a
        // frame writes) are read back right behind their writes, while they still sit in the 256 MiB Infinity Cache:
        // the whole-frame form streams them out to HBM and back.  Measured slower at every group count, and slower
        // still with reduce(g) on a side stream beside bin(g + 1): default 1
        const int ng = g_scatter_groups < num_levels ? g_scatter_groups : num_levels;
        for (int gi = 0; gi < ng; ++gi) {
            const int l0 = (int)((int64_t)num_levels * gi / ng), l1 = (int)((int64_t)num_levels * (gi + 1) / ng);
            launch_bin(l0, l1);
            LNERF_CHECK_LAUNCH("grid_encode_backward(bin)");
            launch_reduce(s, l0, l1);
            LNERF_CHECK_LAUNCH("grid_encode_backward(reduce)");
        }
        return LNERF_OK;
    }
    if (phases & 1) {
        launch_bin(0, num_levels);
        LNERF_CHECK_LAUNCH("grid_encode_backward(bin)");
    }
    if (phases & 2) {
        launch_reduce(s, lv_lo, lv_hi);
        LNERF_CHECK_LAUNCH("grid_encode_backward(reduce)");
    }
    return LNERF_OK;
}

int lnerf_grid_encode_backward(const float *xyzs, float bound, const void *dfeat, int dfeat_dtype, int num_levels,
                               int level_dim, const int32_t *offsets_host, const float *scales_host,
                               const int32_t *res_host, int64_t m_host, const int32_t *m_dev, int64_t level_stride,
                               float *dtable, int variant, void *workspace, size_t workspace_bytes,
                               lnerf_stream_t stream) {
    return scatter_backward(xyzs, bound, dfeat, dfeat_dtype, num_levels, level_dim, offsets_host, scales_host, res_host,
                            m_host, m_dev, level_stride, dtable, variant, workspace, workspace_bytes, stream, nullptr);
}

int lnerf_grid_encode_backward_bf16(const float *xyzs, float bound, const void *dfeat, int dfeat_dtype, int num_levels,
                                    int level_dim, const int32_t *offsets_host, const float *scales_host,
                                    const int32_t *res_host, int64_t m_host, const int32_t *m_dev,
                                    int64_t level_stride, float *dtable_zero, int variant, void *workspace,
                                    size_t workspace_bytes, void *grad_bf16, lnerf_stream_t stream) {
    LNERF_REQUIRE(grad_bf16 && dtable_zero, "grid_encode_backward_bf16: null output");
    LNERF_REQUIRE((((uintptr_t)grad_bf16 | (uintptr_t)dtable_zero) & 15) == 0,
                  "grid_encode_backward_bf16: buffers must be 16-byte aligned");
    FusedUpdate fu;
    memset(&fu, 0, sizeof(fu));
    adam_host_args(fu.a, 0.f, 0.5f, 0.5f, 1.f, 1, nullptr, 1.f, 0);  // (unused in this mode)
    fu.grad_out = (uint16_t *)grad_bf16;
    return scatter_backward(xyzs, bound, dfeat, dfeat_dtype, num_levels, level_dim, offsets_host, scales_host, res_host,
                            m_host, m_dev, level_stride, dtable_zero, variant, workspace, workspace_bytes, stream, &fu);
}

int lnerf_grid_scatter_bin(const float *xyzs, float bound, const void *dfeat, int dfeat_dtype, int num_levels,
                           int level_dim, const int32_t *offsets_host, const float *scales_host, const int32_t *res_host,
                           int64_t m_host, const int32_t *m_dev, int64_t level_stride, float *dtable_zero, int variant,
                           void *workspace, size_t workspace_bytes, lnerf_stream_t stream) {
    LNERF_REQUIRE(dtable_zero, "grid_scatter_bin: null output");
    return scatter_backward(xyzs, bound, dfeat, dfeat_dtype, num_levels, level_dim, offsets_host, scales_host, res_host,
                            m_host, m_dev, level_stride, dtable_zero, variant, workspace, workspace_bytes, stream, nullptr,
                            1);
}

int lnerf_grid_scatter_reduce_bf16(float bound, int num_levels, int level_dim, const int32_t *offsets_host,
                                   const float *scales_host, const int32_t *res_host, int64_t m_host,
                                   int64_t level_stride, int level_lo, int level_hi, float *dtable_zero, int variant,
                                   void *workspace, size_t workspace_bytes, void *grad_bf16, lnerf_stream_t stream) {
    LNERF_REQUIRE(grad_bf16 && dtable_zero, "grid_scatter_reduce_bf16: null output");
    LNERF_REQUIRE((((uintptr_t)grad_bf16 | (uintptr_t)dtable_zero) & 15) == 0,
                  "grid_scatter_reduce_bf16: buffers must be 16-byte aligned");
    FusedUpdate fu;
    memset(&fu, 0, sizeof(fu));
    adam_host_args(fu.a, 0.f, 0.5f, 0.5f, 1.f, 1, nullptr, 1.f, 0);  // (unused in this mode)
    fu.grad_out = (uint16_t *)grad_bf16;
    return scatter_backward(nullptr, bound, nullptr, LNERF_F32, num_levels, level_dim, offsets_host, scales_host, res_host,
                            m_host, nullptr, level_stride, dtable_zero, variant, workspace, workspace_bytes, stream, &fu, 2,
                            level_lo, level_hi);
}

int lnerf_grid_encode_backward_adam(const float *xyzs, float bound, const void *dfeat, int dfeat_dtype, int num_levels,
                                    int level_dim, const int32_t *offsets_host, const float *scales_host,
                                    const int32_t *res_host, int64_t m_host, const int32_t *m_dev,
                                    int64_t level_stride, float *dtable_zero, int variant, void *workspace,
                                    size_t workspace_bytes, float *table, float *exp_avg, float *exp_avg_sq,
                                    void *shadow_bf16, float lr, float beta1, float beta2, float eps, int step,
                                    const int32_t *step_dev, float grad_scale, lnerf_stream_t stream) {
    LNERF_REQUIRE(table && exp_avg && exp_avg_sq, "grid_encode_backward_adam: null optimiser state");
    LNERF_REQUIRE(step_dev || step >= 1, "grid_encode_backward_adam: step must be >= 1 (got %d)", step);
    LNERF_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f,
                  "grid_encode_backward_adam: betas must be in [0,1)");
    LNERF_REQUIRE((((uintptr_t)table | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)dtable_zero) & 15) == 0,
                  "grid_encode_backward_adam: buffers must be 16-byte aligned");
    LNERF_REQUIRE(!shadow_bf16 || ((uintptr_t)shadow_bf16 & 7) == 0, "grid_encode_backward_adam: shadow must be 8-byte aligned");
    FusedUpdate fu;
    fu.p = table; fu.m = exp_avg; fu.v = exp_avg_sq; fu.shadow = (uint16_t *)shadow_bf16;
    fu.grad_out = nullptr;
    adam_host_args(fu.a, lr, beta1, beta2, eps, step, step_dev, grad_scale, 0);
    return scatter_backward(xyzs, bound, dfeat, dfeat_dtype, num_levels, level_dim, offsets_host, scales_host, res_host,
                            m_host, m_dev, level_stride, dtable_zero, variant, workspace, workspace_bytes, stream, &fu);
}

// the MLP half of a step tail: argument checks + the kernel-side descriptor
static int fill_slab_adam(const char *who, SlabAdam &sa, const void *mlp_workspace, size_t mlp_workspace_bytes,
                          int mlp_precision, int out_dim, int64_t m_host, float *const *params_host,
                          float *const *exp_avg_host, float *const *exp_avg_sq_host, float mlp_lr,
                          const int32_t *const *maps_host) {
    memset(&sa, 0, sizeof(sa));
    LNERF_REQUIRE(out_dim >= 2 && out_dim <= 8, "%s: out_dim must be in [2,8]", who);
    LNERF_REQUIRE(mlp_precision == LNERF_F32 || mlp_precision == LNERF_BF16, "%s: bad precision tag", who);
    LNERF_REQUIRE(mlp_workspace && mlp_workspace_bytes >= lnerf_mlp_backward_workspace_bytes(out_dim),
                  "%s: MLP workspace too small", who);
    LNERF_REQUIRE(params_host && exp_avg_host && exp_avg_sq_host && m_host > 0, "%s: null MLP state", who);
    for (int k = 0; k < 6; ++k) {
        LNERF_REQUIRE(params_host[k] && exp_avg_host[k] && exp_avg_sq_host[k], "%s: null MLP tensor %d", who, k);
        sa.p[k] = params_host[k]; sa.m[k] = exp_avg_host[k]; sa.v[k] = exp_avg_sq_host[k];
    }
    if (maps_host) {
        for (int k = 0; k < 3; ++k) {
            LNERF_REQUIRE(maps_host[k] && ((uintptr_t)maps_host[k] & 7) == 0, "%s: bad fragment map %d", who, k);
            sa.map[k] = maps_host[k];
        }
        sa.shadow = (uint16_t *)const_cast<void *>(mlp_workspace);   // the fragment image heads the workspace
    }
    sa.slabs = reinterpret_cast<const float *>(static_cast<const char *>(mlp_workspace) + MLP_FRAG_BYTES);
    sa.n_slabs = lnerf_mlp_backward_slabs(m_host, mlp_precision);
    sa.out_dim = out_dim;
    sa.lr = mlp_lr;
    return LNERF_OK;
}

int lnerf_grid_encode_backward_adam_tail(const float *xyzs, float bound, const void *dfeat, int dfeat_dtype, int num_levels,
                                         int level_dim, const int32_t *offsets_host, const float *scales_host,
                                         const int32_t *res_host, int64_t m_host, const int32_t *m_dev,
                                         int64_t level_stride, float *dtable_zero, int variant, void *workspace,
                                         size_t workspace_bytes, float *table, float *exp_avg, float *exp_avg_sq,
                                         void *shadow_bf16, float lr, const void *mlp_workspace,
                                         size_t mlp_workspace_bytes, int mlp_precision, int out_dim,
                                         float *const *params_host, float *const *exp_avg_host,
                                         float *const *exp_avg_sq_host, float mlp_lr, const int32_t *const *maps_host,
                                         float beta1, float beta2, float eps, int step, int32_t *step_dev, float grad_scale,
                                         int flags, lnerf_stream_t stream) {
    LNERF_REQUIRE(table && exp_avg && exp_avg_sq, "grid_encode_backward_adam_tail: null optimiser state");
    LNERF_REQUIRE(step_dev, "grid_encode_backward_adam_tail: needs the device counter pair (int32[2])");
    LNERF_REQUIRE(m_host > 0, "grid_encode_backward_adam_tail: needs m_host > 0 (use lnerf_step_tail for an empty frame)");
    LNERF_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f,
                  "grid_encode_backward_adam_tail: betas must be in [0,1)");
    LNERF_REQUIRE((((uintptr_t)table | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)dtable_zero) & 15) == 0,
                  "grid_encode_backward_adam_tail: buffers must be 16-byte aligned");
    LNERF_REQUIRE(!shadow_bf16 || ((uintptr_t)shadow_bf16 & 7) == 0,
                  "grid_encode_backward_adam_tail: shadow must be 8-byte aligned");
    FusedUpdate fu;
    fu.p = table; fu.m = exp_avg; fu.v = exp_avg_sq; fu.shadow = (uint16_t *)shadow_bf16;
    fu.grad_out = nullptr;
    adam_host_args(fu.a, lr, beta1, beta2, eps, step, step_dev, grad_scale, 0);
    TailJob tj;
    memset(&tj, 0, sizeof(tj));
    if (mlp_workspace) {
        int rc = fill_slab_adam("grid_encode_backward_adam_tail", tj.sa, mlp_workspace, mlp_workspace_bytes, mlp_precision,
                                out_dim, m_host, params_host, exp_avg_host, exp_avg_sq_host, mlp_lr, maps_host);
        if (rc) return rc;
    }
    tj.tick = step_dev;
    tj.do_tick = (flags & LNERF_TAIL_TICK) ? 1 : 0;
    tj.clear_gmax = (flags & LNERF_TAIL_CLEAR_SCATTER) ? 1 : 0;
    return scatter_backward(xyzs, bound, dfeat, dfeat_dtype, num_levels, level_dim, offsets_host, scales_host, res_host,
                            m_host, m_dev, level_stride, dtable_zero, variant, workspace, workspace_bytes, stream, &fu, 3, 0,
                            -1, &tj);
}

int lnerf_step_tail(int num_levels, int level_dim, const int32_t *offsets_host, const float *scales_host,
                    const int32_t *res_host, int64_t m_host, int variant, void *scatter_workspace,
                    size_t scatter_workspace_bytes, float *dtable_zero, float *table, float *exp_avg, float *exp_avg_sq,
                    void *shadow_bf16, float table_lr, const void *mlp_workspace, size_t mlp_workspace_bytes,
                    int mlp_precision, int out_dim, float *const *params_host, float *const *exp_avg_host,
                    float *const *exp_avg_sq_host, float mlp_lr, const int32_t *const *maps_host, float beta1, float beta2,
                    float eps, int step, int32_t *step_dev, float grad_scale, int flags, lnerf_stream_t stream) {
    const bool with_scatter = num_levels > 0, with_mlp = mlp_workspace != nullptr;
    LNERF_REQUIRE(with_scatter || with_mlp, "step_tail: nothing to do");
    LNERF_REQUIRE(step_dev || step >= 1, "step_tail: step must be >= 1 (got %d)", step);
    LNERF_REQUIRE(!(flags & (LNERF_TAIL_TICK | LNERF_TAIL_CLEAR_SCATTER)) || step_dev,
                  "step_tail: the tick / the clearing epilogue need the device counter pair (int32[2])");
    LNERF_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "step_tail: betas must be in [0,1)");
    GridMeta meta;
    BucketMeta bm;
    ScatterPlan plan;
    memset(&meta, 0, sizeof(meta));
    memset(&bm, 0, sizeof(bm));
    memset(&plan, 0, sizeof(plan));
    FusedUpdate fu;
    memset(&fu, 0, sizeof(fu));
    adam_host_args(fu.a, table_lr, beta1, beta2, eps, step, step_dev, grad_scale, 0);
    unsigned int *gmax = nullptr;
    int32_t *arrive = nullptr;
    const bool packed = (variant & 0xFF) == 3;
    if (with_scatter) {
        int rc = fill_meta("step_tail", meta, num_levels, level_dim, offsets_host, scales_host, res_host);
        if (rc) return rc;
        LNERF_REQUIRE(((variant & 0xFF) == 2 || packed) && m_host > 0, "step_tail: needs scatter variant 2 / 3 and m_host > 0");
        LNERF_REQUIRE(fill_bucket_meta(meta, m_host, bm, plan) == 0, "step_tail: level too large for the bucketed scatter");
        LNERF_REQUIRE(scatter_workspace && scatter_workspace_bytes >= plan.total(), "step_tail: scatter workspace too small");
        LNERF_REQUIRE(table && exp_avg && exp_avg_sq && dtable_zero, "step_tail: null optimiser state");
        LNERF_REQUIRE((((uintptr_t)table | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq | (uintptr_t)scatter_workspace) & 15) == 0,
                      "step_tail: buffers must be 16-byte aligned");
        char *wsb = (char *)scatter_workspace;
        gmax = (unsigned int *)wsb;
        arrive = (int32_t *)(wsb + HDR_ARRIVE_OFF);
        fu.p = table; fu.m = exp_avg; fu.v = exp_avg_sq; fu.shadow = (uint16_t *)shadow_bf16;
    }
    LNERF_REQUIRE(!(flags & (LNERF_TAIL_CLEAR_SCATTER | LNERF_TAIL_TICK)) || with_scatter,
                  "step_tail: the tick / the clearing epilogue keep their arrival counters in the scatter workspace");
    SlabAdam sa;
    memset(&sa, 0, sizeof(sa));
    int n_slab_blocks = 0;
    if (with_mlp) {
        int rc = fill_slab_adam("step_tail", sa, mlp_workspace, mlp_workspace_bytes, mlp_precision, out_dim, m_host,
                                params_host, exp_avg_host, exp_avg_sq_host, mlp_lr, maps_host);
        if (rc) return rc;
        n_slab_blocks = (int)div_up(MLP_SLAB, TAIL_P);
    }
    const int do_tick = (flags & LNERF_TAIL_TICK) ? 1 : 0, clr = (flags & LNERF_TAIL_CLEAR_SCATTER) ? 1 : 0;
    // no MLP: the tick / the clearing epilogue still run -- ONE block that arrives on its own (k_step_tail's
    // `sa.slabs == nullptr` branch); a call with nothing at all to do returns without a launch
    if (n_slab_blocks == 0 && (do_tick || clr)) n_slab_blocks = 1;
    const dim3 g((unsigned)n_slab_blocks);
    if (g.x == 0) return LNERF_OK;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_step_tail, g, dim3(256), 0, s, gmax, fu.a, sa, step_dev, arrive, do_tick, clr);
    LNERF_CHECK_LAUNCH("step_tail");
    return LNERF_OK;
}

}  // extern "C"
