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
// VALU) are shared by all corners ((y+1)*P == y*P + P mod 2^32), dense levels add strides to one base
__device__ __forceinline__ void corner_rows(uint32_t gx, uint32_t gy, uint32_t gz, uint32_t res, uint32_t hsize,
                                            uint32_t row[8]) {
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
        corner_rows(p.gx, p.gy, p.gz, res, hsize, rows);
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
        corner_rows(p.gx, p.gy, p.gz, res, hsize, rows);
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
// Backward, variant 2: two-pass bucketed scatter (no global float atomics on the hot path).
//
// Scattered 8-byte float atomics run at the memory side at ~20 G requests/s chip-wide
// (MI355X_MICROARCH.md "Global float atomics"): 55 M vertex updates per frame cost ~10 ms that
// way.  Instead every level's table is cut into buckets of BK_ROWS consecutive rows (64 KiB of
// f32x2 accumulators = one LDS tile):
//   pass 1 (k_scatter_bin)    one thread per (sample, level) computes its 8 (row, w*g) records (runs of
//                             samples in one cell merged first on coarse levels), groups the workgroup's
//                             records by bucket in LDS, reserves a span per touched bucket with ONE returning
//                             global atomic, and appends the records to the bucket's region with plain stores;
//   pass 2 (k_scatter_reduce) one workgroup per (bucket, slice) streams its records (coalesced
//                             16 B/lane), accumulates them with LDS float atomics and adds the
//                             64 KiB tile to the gradient table with coalesced stores.
// A bucket that overflows its region falls back to global atomics for the excess records, so the
// result is always complete.
constexpr int BK_SHIFT = 12, BK_ROWS = 1 << BK_SHIFT;  // 4096 rows * 8 B = 32 KiB of accumulators
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
static_assert(BK_SHIFT == 12, "Rec8 stores 12 row bits");

// Distance between two buckets' cursors (and two levels' maxima) in int32 words: one 128-byte line each.  Device-scope
// atomics execute at the memory side, and those that hit ONE line are served one after the other whatever word they
// name: with the cursors packed (32 to a line, a level's 128 on 4 lines) the binning pass's 1.3 M reservations per
// launch queued on 64 lines -- same-box A/B of the bench step: scatter 0.274 -> 0.254 ms with a line per cursor
// (64-byte spacing: no change; 256-byte: same as 128).
#ifndef LNERF_CUR_STRIDE
#define LNERF_CUR_STRIDE 32
#endif
constexpr int CUR_STRIDE = LNERF_CUR_STRIDE;

struct BucketMeta {
    int nb[LNERF_MAX_LEVELS];            // buckets per level
    int bstart[LNERF_MAX_LEVELS + 1];    // first global bucket id of the level
    int cap[LNERF_MAX_LEVELS];           // record capacity of each bucket of the level
    int slices[LNERF_MAX_LEVELS];        // pass-2 workgroups per bucket
    int compact[LNERF_MAX_LEVELS];       // 1: merge runs of equal rows inside a wavefront before binning
    int wgstart[LNERF_MAX_LEVELS + 1];   // first pass-2 workgroup of the level
    long long rstart[LNERF_MAX_LEVELS];  // first record slot of the level's region
    int pstart[LNERF_MAX_LEVELS];        // sliced levels: first partial-sum tile of the level (pass 2 -> finish)
    int fstart[LNERF_MAX_LEVELS];        // sliced levels: first bucket index in the finishing pass's grid
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
// tile-major (tile, level) list, the level rotated by one per round (every workgroup sees every level; the levels'
// bucket cursors are hit evenly), and fetch the next item's inputs while the current one is processed.  rocprofv3
// (profiles/r02_pmc_scatter*.json) shows the pass bound by vector-ALU issue -- ~590 VALU instructions per wavefront
// and item in round 1's form, 73 % of the SIMD cycles -- and by the per-item chain of LDS / global round trips, not by
// bytes.  Hence, in this form:
//   * hashed levels (rows of a wavefront spread over all buckets): records ranked with LDS counters, grouped by
//     bucket in an LDS stage (exact packing from a count scan), ONE returning global atomic per touched bucket
//     reserves the span, coalesced copy-out; two barriers per item (counters double-buffered);
//   * dense levels (rows = x + y s + z s^2: a wavefront of consecutive samples touches a handful of buckets): no
//     stage, no count scan and no workgroup barrier -- every wavefront ranks its records with a private LDS
//     histogram, reserves its own spans and stores the records directly (consecutive ranks = consecutive slots);
//   * the level's largest |value| (fixed-point scale of pass 2) is bounded from |g| (weights <= 1, runs <= 64
//     samples) instead of being measured on the 16 products, one LDS maximum per level and workgroup; records are
//     packed with bit-field inserts; run sums use fused DPP adds.
constexpr int BIN_T = 512;                 // threads per workgroup = samples per item
constexpr int BIN_WAVES = BIN_T / 64;
// dense levels with this many buckets take the direct path (fewer: every wavefront's reservation would hit the same
// one or two cursor words -- one word takes ~88 returning atomics per microsecond; more: the per-wave histogram)
constexpr int BIN_DIRECT_MIN = 8, BIN_DIRECT_NB = 64;

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
    uint32_t res, hsize, off;
    int level, nb, cap, b0;
    long long rstart;
    bool compact, direct;
};
#define LNERF_BIN_LEVEL(lv)                                                                                          \
    BinLevel {                                                                                                       \
        meta.scales[lv], (uint32_t)meta.res[lv], (uint32_t)(meta.offsets[(lv) + 1] - meta.offsets[lv]),              \
            (uint32_t)meta.offsets[lv], (lv), bm.nb[lv], bm.cap[lv], bm.bstart[lv], bm.rstart[lv], bm.compact[lv] != 0, \
            (uint64_t)(meta.res[lv] + 1) * (meta.res[lv] + 1) * (meta.res[lv] + 1) <=                                 \
                    (uint64_t)(meta.offsets[(lv) + 1] - meta.offsets[lv]) &&                                          \
                bm.nb[lv] >= BIN_DIRECT_MIN && bm.nb[lv] <= BIN_DIRECT_NB                                             \
    }

template <typename REC>
__global__ void __launch_bounds__(BIN_T, (sizeof(REC) == 8 ? 6 : 4))
k_scatter_bin(const float *__restrict__ xyzs, float bound, const float *__restrict__ dfeat, GridMeta meta, BucketMeta bm,
              int64_t m_host, const int32_t *__restrict__ m_dev, int64_t level_stride, int32_t *__restrict__ cursor,
              unsigned int *__restrict__ gmax, REC *__restrict__ recs, float *__restrict__ dtable, int skip_zero) {
    __shared__ int s_cnt[2][BK_MAX_PER_LEVEL];  // records of the item per bucket (two sets: hashed items alternate)
    __shared__ int s_base[BK_MAX_PER_LEVEL];    // first slot reserved in the bucket's global region
    __shared__ int s_off[BK_MAX_PER_LEVEL];     // first slot of the bucket in the LDS stage
    __shared__ int s_dest[BK_MAX_PER_LEVEL];    // global slot of the bucket's first staged record, minus its stage offset
    __shared__ int s_ovf[2];                    // some bucket of this item ran past its region
    __shared__ REC s_stage[BIN_T * 8];          // the item's records, grouped by bucket (48 KiB, 32 KiB packed)
    __shared__ uint8_t s_bkt[REC::kPacked ? BIN_T * 8 : 4];  // packed records do not name their bucket: kept beside
    __shared__ int s_wave[BIN_WAVES][BIN_DIRECT_NB];          // dense levels: per-wave histogram, then per-wave bases
    __shared__ unsigned int s_lmax[LNERF_MAX_LEVELS];         // per level: bound of |value| seen by this workgroup
    int32_t M = (int32_t)m_host;
    if (m_dev) { const int32_t md = *m_dev; M = md < M ? md : M; }
    const int L = meta.num_levels;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // item k of this workgroup: tile t0 + k * tstep, level (l0 + k) mod L   (gridDim.x is a multiple of L)
    const int tstep = gridDim.x / L;
    const float two_b = 2.0f * bound;
    const bool pow2_bound = (__float_as_uint(two_b) & 0x007FFFFFu) == 0u;
    // (exact when the bound is a power of two, the only case it is used in; wave-uniform, kept in a scalar register)
    float inv_two_b;
    asm("v_readfirstlane_b32 %0, %1" : "=s"(inv_two_b) : "v"(1.0f / two_b));
    for (int i = tid; i < 2 * BK_MAX_PER_LEVEL; i += BIN_T) (&s_cnt[0][0])[i] = 0;
    if (tid < LNERF_MAX_LEVELS) s_lmax[tid] = 0u;
    if (tid < 2) s_ovf[tid] = 0;
    // ---- inputs of an item (5 dwords per lane), fetched while the previous item is processed (the pass waits on
    // memory round trips, not on bytes)
    float n_x = 0.f, n_y = 0.f, n_z = 0.f;
    float2 n_g = make_float2(0.f, 0.f);
    auto fetch = [&](int lv, int tl) __attribute__((always_inline)) {
        const int mm = tl * BIN_T + tid;
        n_x = n_y = n_z = 0.f;
        n_g = make_float2(0.f, 0.f);
        if (mm < M) {
            n_g = reinterpret_cast<const float2 *>(dfeat)[(int64_t)lv * level_stride + mm];
            n_x = xyzs[(int64_t)mm * 3]; n_y = xyzs[(int64_t)mm * 3 + 1]; n_z = xyzs[(int64_t)mm * 3 + 2];
        }
    };
    // a record that cannot be placed in its bucket's region: finished with global float atomics
    auto spill = [&](uint32_t level_off, const REC &r, int b) __attribute__((always_inline)) {
        float *lt = dtable + (int64_t)level_off * 2;
        const int64_t full_row = ((int64_t)b << BK_SHIFT) | r.row_in_bucket();
        atomicAdd(lt + full_row * 2, r.a());
        atomicAdd(lt + full_row * 2 + 1, r.b());
    };
    int l = (int)(blockIdx.x % L);
    int tile = (int)(blockIdx.x / L);
    bool have = tile * BIN_T < M;
    if (have) fetch(l, tile);
    BIN_STAMP_INIT();
    __syncthreads();
    int hk = 0;  // hashed items so far (selects the counter set)
    while (have) {
        const BinLevel lv = LNERF_BIN_LEVEL(l);
        const int nb = lv.nb, cap = lv.cap;
        REC *lrec = recs + lv.rstart;
        const int l_next = l + 1 == L ? 0 : l + 1;
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
            corner_rows(p.gx, p.gy, p.gz, lv.res, lv.hsize, row);
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
        // ---- D (a lambda: placed behind the reservations on hashed levels): the values w * g (run sums on coarse
        // levels), packed into records; the bound of |value| goes to the level's LDS maximum
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
        // two buckets (dense levels, tiny tables) one LDS atomic per (wave, bucket) instead of one per lane -- same-address
        // LDS atomics serialise
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
        // the rank of a record inside its bucket -- among the item's records (hashed) / the wavefront's (dense), < 4096 --
        // is kept in bits [20, 32) of its row (rows of a level are < 2^20: at most 256 buckets of 4096 rows)
        constexpr uint32_t ROW_MASK = (1u << 20) - 1u;
        int my_base = 0;   // hashed: thread b holds the reserved base of bucket b; dense: lane b the wave's span base
        if (lv.direct) {
            // ================= dense level: every wavefront places its own records, no workgroup barrier ==========
            int *hist = s_wave[wave];
            if (lane < BIN_DIRECT_NB) hist[lane] = 0;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (LDS is in order within a wave)
            if (lv.compact) {  // few lanes emit (run tails): plain per-lane counter increments (measured 17 us faster)
                if (emit) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) row[c] |= (uint32_t)atomicAdd(&hist[row[c] >> BK_SHIFT], 1) << 20;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) rank_by_ballot(hist, c);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane < nb) {
                const int n = hist[lane];
                if (n) my_base = atomicAdd(&cursor[(lv.b0 + lane) * CUR_STRIDE], n);  // the wave's span in bucket `lane`
            }
        } else {
            // ================= hashed level ==================================================================
            ++hk;
            // ---- B: rank every record inside its bucket (item-local)
            if (nb <= 32 && !lv.compact) {  // wave-uniform: every lane emits into one or two buckets
#pragma unroll
                for (int c = 0; c < 8; ++c) rank_by_ballot(s_cnt[cur], c);
            } else if (emit) {
#pragma unroll
                for (int c = 0; c < 8; ++c) row[c] |= (uint32_t)atomicAdd(&s_cnt[cur][row[c] >> BK_SHIFT], 1) << 20;
            }
            BIN_STAMP(2);
            __syncthreads();  // barrier 1: the item's bucket counts are final
            BIN_STAMP(3);
            // (the other set was last read before barrier 3 of the previous hashed item: clear it for the next one)
            for (int i = tid; i < BK_MAX_PER_LEVEL; i += BIN_T) s_cnt[cur ^ 1][i] = 0;
            if (tid == 0) s_ovf[cur ^ 1] = 0;
            // ---- C: ONE returning global atomic per touched bucket reserves its span
            if (tid < nb) {
                const int c = s_cnt[cur][tid];
                if (c) my_base = atomicAdd(&cursor[(lv.b0 + tid) * CUR_STRIDE], c);
            }
        }
        // the next item's inputs follow the reservations into the memory queue; both are consumed after the arithmetic
        if (have_next) fetch(l_next, tile_next);
        values();
        BIN_STAMP(4);
        if (lv.direct) {
            int *wbase = s_wave[wave];
            if (lane < BIN_DIRECT_NB) wbase[lane] = my_base;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (emit) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const int b = (int)((row[c] & ROW_MASK) >> BK_SHIFT);
                    const int pos = wbase[b] + (int)(row[c] >> 20);
                    if (pos < cap) lrec[(int64_t)b * cap + pos] = rec[c];
                    else spill(lv.off, rec[c], b);
                }
            }
            BIN_STAMP(8);
            l = l_next; tile = tile_next; have = have_next;
            continue;
        }
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
        // ---- F: group the records by bucket in LDS
        if (emit) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int b = (int)((row[c] & ROW_MASK) >> BK_SHIFT);
                const int slot = s_off[b] + (int)(row[c] >> 20);
                s_stage[slot] = rec[c];
                if (REC::kPacked) s_bkt[slot] = (uint8_t)b;
            }
        }
        BIN_STAMP(5);
        if (tid < nb) {  // waits for the reservations
            s_base[tid] = my_base;
            s_dest[tid] = tid * cap + my_base - s_off[tid];
            if (my_base + s_cnt[cur][tid] > cap) s_ovf[cur] = 1;
        }
        BIN_STAMP(6);
        __syncthreads();  // barrier 3: stage and destinations complete
        BIN_STAMP(7);
        // the prefetched inputs are pinned in registers here, so that the next item starts without waiting for the
        // stores below to be acknowledged (one in-order memory counter covers loads and stores)
        asm volatile("" : "+v"(n_g.x), "+v"(n_g.y), "+v"(n_x), "+v"(n_y), "+v"(n_z));
        // ---- G: copy out: consecutive lanes -> consecutive slots of (mostly) the same bucket: coalesced
        auto bucket_of = [&](const REC &r, int i) __attribute__((always_inline)) -> int {
            if constexpr (REC::kPacked) return (int)s_bkt[i];
            else return (int)(r.row >> BK_SHIFT);
        };
        if (!s_ovf[cur]) {  // uniform fast path: every record of the item has a slot
            char *lbytes = reinterpret_cast<char *>(lrec);  // (a level's region is < 4 GiB: 32-bit byte offsets)
            for (int i = tid; i < total; i += BIN_T) {
                const REC r = s_stage[i];
                const uint32_t at = (uint32_t)(s_dest[bucket_of(r, i)] + i) * (uint32_t)sizeof(REC);
                *reinterpret_cast<REC *>(lbytes + at) = r;
            }
        } else {
            for (int i = tid; i < total; i += BIN_T) {
                const REC r = s_stage[i];
                const int b = bucket_of(r, i);
                const int slot = s_base[b] + (i - s_off[b]);
                if (slot < cap) lrec[(int64_t)b * cap + slot] = r;
                else spill(lv.off, r, b);   // bucket region full: finish this record with global atomics
            }
        }
        BIN_STAMP(8);
        // (the next hashed item rewrites s_off / s_stage / s_dest only after ITS barrier 1, which every wave reaches
        // after finishing the copy-out above)
        l = l_next; tile = tile_next; have = have_next;
    }
    __syncthreads();
    if (tid < L && s_lmax[tid] != 0u) atomicMax(&gmax[tid * CUR_STRIDE], s_lmax[tid]);  // one value per LEVEL and workgroup
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
template <int BITS>
__device__ __forceinline__ FixScale fix_scale(unsigned int gmax_bits) {
    int e = (int)(gmax_bits >> 23);
    e = e < 1 ? 1 : (e > 254 ? 254 : e);
    int k = BITS + 126 - e;
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
// as a tile of `partials`, and k_scatter_finish adds the tiles up -- integer addition, so the result does not depend
// on how many slices there were or in which order they ran: the whole gradient is bitwise reproducible.
template <int RT, typename REC, bool FUSE>
__global__ void __launch_bounds__(RT)
k_scatter_reduce(GridMeta meta, BucketMeta bm, const int32_t *__restrict__ cursor,
                 const unsigned int *__restrict__ gmax, const REC *__restrict__ recs, float *__restrict__ dtable,
                 long long *__restrict__ partials, int wg_lo, FusedUpdate fu) {
    __shared__ long long acc[BK_ROWS * 2];  // [feature][row]: a wave's 64 random rows spread over 32 bank pairs
    // locate (level, bucket, slice) of this workgroup
    const int wg = (int)blockIdx.x + wg_lo;
    int l = 0;
    while (l + 1 < meta.num_levels && wg >= bm.wgstart[l + 1]) ++l;
    const int Smax = bm.slices[l];
    const int local = wg - bm.wgstart[l];
    const int b = local / Smax, s = local - b * Smax;
    const int cap = bm.cap[l];
    const int n_raw = cursor[(bm.bstart[l] + b) * CUR_STRIDE];  // > cap: the excess records went to dtable with global atomics
    const int n = n_raw < cap ? n_raw : cap;
    const int S = active_slices(n, Smax);
    if (s >= S) return;    // uniform per workgroup
    const bool direct = S == 1;        // this workgroup sums the whole bucket: it finishes the rows itself
    const bool fuse = FUSE && direct;
    const int lo = (int)(((long long)n * s) / S), hi = (int)(((long long)n * (s + 1)) / S);
    const bool have = hi > lo;  // uniform
    if (!have && !fuse) return;  // (a fused bucket without records still owes its rows the Adam step, g = 0)
    constexpr int FB = FixBits<REC>::kBits;
    const FixScale fs = fix_scale<FB>(gmax[l * CUR_STRIDE]);  // from the bound of |value| of the LEVEL (found by pass 1)
    const int tid = threadIdx.x;
    RED_STAMP_INIT();
    const int hsize = meta.offsets[l + 1] - meta.offsets[l];
    const int row0 = b << BK_SHIFT;
    int rows = hsize - row0;
    rows = rows < BK_ROWS ? rows : BK_ROWS;
    const int64_t R0 = (int64_t)meta.offsets[l] + row0;
    if (have) {
        for (int i = tid; i < BK_ROWS * 2; i += RT) acc[i] = 0ll;
        __syncthreads();
        RED_STAMP(10);
        const REC *rp = recs + bm.rstart[l] + (long long)b * cap;
        unsigned long long *ua = reinterpret_cast<unsigned long long *>(acc);
        auto add = [&](const REC &r) {
            const uint32_t a0 = r.row_in_bucket();
            atomicAdd(&ua[a0], (unsigned long long)to_fixed<FB>((r.a() * fs.sc_a) * fs.sc_b));
            atomicAdd(&ua[a0 + BK_ROWS], (unsigned long long)to_fixed<FB>((r.b() * fs.sc_a) * fs.sc_b));
        };
        // the pass waits on its record loads (rocprofv3: 82 % of wave cycles parked): keep four loads in flight
        // per lane
        int i = lo + tid;
        for (; i + 3 * RT < hi; i += 4 * RT) {
            const REC r0 = rp[i], r1 = rp[i + RT], r2 = rp[i + 2 * RT], r3 = rp[i + 3 * RT];
            add(r0); add(r1); add(r2); add(r3);
        }
        for (; i < hi; i += RT) add(rp[i]);
        RED_STAMP(11);
        __syncthreads();
        RED_STAMP(12);
    }
    if (!direct) {  // sliced bucket: hand the exact sums to k_scatter_finish
        long long *pt = partials + ((int64_t)bm.pstart[l] + (int64_t)b * Smax + s) * (BK_ROWS * 2);
        for (int i = tid; i < BK_ROWS * 2; i += RT) pt[i] = acc[i];
        return;
    }
    float *dst = dtable + R0 * 2;
    if (fuse) {
        AdamArgs a = fu.a;
        adam_bias(a);
        a.zero_grad = 0;
        const bool ovf = n_raw > cap;  // uniform
        float2 *p2 = reinterpret_cast<float2 *>(fu.p) + R0, *m2 = reinterpret_cast<float2 *>(fu.m) + R0;
        float2 *v2 = reinterpret_cast<float2 *>(fu.v) + R0;
        uint32_t *sh = fu.shadow ? reinterpret_cast<uint32_t *>(fu.shadow) + R0 : nullptr;
        auto grad_of = [&](int r, float &g0, float &g1) {
            g0 = 0.f; g1 = 0.f;
            if (have) {
                g0 = ((float)acc[r] * fs.un_a) * fs.un_b;
                g1 = ((float)acc[r + BK_ROWS] * fs.un_a) * fs.un_b;
            }
            if (ovf) {  // what pass 1 could not place: consume it and leave dtable zero again
                const float2 d = reinterpret_cast<float2 *>(dst)[r];
                g0 = d.x + g0;
                g1 = d.y + g1;
                reinterpret_cast<float2 *>(dst)[r] = make_float2(0.f, 0.f);
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
        // usual case (full bucket, even first row): two rows per lane and access (16 B).  (Fetching the lane's
        // parameters and moments ahead of the record stream was measured 35 us SLOWER.)
        constexpr int NQ = (BK_ROWS / 2 + RT - 1) / RT;  // row pairs per lane
        if (((R0 | rows) & 1) == 0 && rows == BK_ROWS && (BK_ROWS / 2) % RT == 0) {
            float4 *p4 = reinterpret_cast<float4 *>(p2), *m4 = reinterpret_cast<float4 *>(m2);
            float4 *v4 = reinterpret_cast<float4 *>(v2);
            uint2 *sh2 = reinterpret_cast<uint2 *>(sh);
            float4 P[NQ], Mv[NQ], V[NQ];
#pragma unroll
            for (int j = 0; j < NQ; ++j) {  // all of the lane's loads first: six 16-byte loads in flight
                const int q = tid + j * RT;
                P[j] = p4[q]; Mv[j] = m4[q]; V[j] = v4[q];
            }
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
                p4[q] = P[j]; m4[q] = Mv[j]; v4[q] = V[j];
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
            float2 P = p2[r], Mv = m2[r], V = v2[r];
            float g0, g1;
            grad_of(r, g0, g1);
            adam_one(P.x, g0, Mv.x, V.x, a);
            adam_one(P.y, g1, Mv.y, V.y, a);
            p2[r] = P; m2[r] = Mv; v2[r] = V;
            if (sh) sh[r] = (uint32_t)f32_to_bf16(P.x) | ((uint32_t)f32_to_bf16(P.y) << 16);
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

// Finishing pass of the sliced levels: one thread per table row adds the active slices' exact partial sums
// (k_scatter_reduce), converts once and adds the result to dtable -- or applies the Adam step (FUSE).
template <bool FUSE, int FB>
__global__ void __launch_bounds__(256)
k_scatter_finish(GridMeta meta, BucketMeta bm, const int32_t *__restrict__ cursor, const unsigned int *__restrict__ gmax,
                 const long long *__restrict__ partials, float *__restrict__ dtable, FusedUpdate fu, int lv_lo, int lv_hi) {
    constexpr int WG_PER_BUCKET = BK_ROWS / 256;
    const int fb = blockIdx.x / WG_PER_BUCKET;  // index among the buckets of sliced levels
    int l = 0;
    while (l + 1 < meta.num_levels && (bm.slices[l] <= 1 || fb >= bm.fstart[l] + bm.nb[l])) ++l;
    if (bm.slices[l] <= 1) return;  // (cannot happen: the grid covers sliced buckets only)
    if (l < lv_lo || l >= lv_hi) return;  // a launch over a level range (pipelined data-parallel exchange)
    const int b = fb - bm.fstart[l];
    const int Smax = bm.slices[l], cap = bm.cap[l];
    const int n_raw = cursor[(bm.bstart[l] + b) * CUR_STRIDE];
    const int n = n_raw < cap ? n_raw : cap;
    const int S = active_slices(n, Smax);
    if (S <= 1) return;  // the bucket was finished by its single pass-2 workgroup
    const int r = (blockIdx.x % WG_PER_BUCKET) * 256 + threadIdx.x;
    const int hsize = meta.offsets[l + 1] - meta.offsets[l];
    const int row0 = b << BK_SHIFT;
    if (row0 + r >= hsize) return;
    const long long *pt = partials + ((int64_t)bm.pstart[l] + (int64_t)b * Smax) * (BK_ROWS * 2);
    long long q0 = 0ll, q1 = 0ll;
    int s = 0;
    for (; s + 4 <= S; s += 4) {   // eight loads in flight per lane: the slice count is dynamic, an un-unrolled loop pays
                                   // one memory round trip per slice (integer sums: the order is free)
        long long a[4], b[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[k] = pt[(int64_t)(s + k) * (BK_ROWS * 2) + r];
            b[k] = pt[(int64_t)(s + k) * (BK_ROWS * 2) + BK_ROWS + r];
        }
        q0 += (a[0] + a[1]) + (a[2] + a[3]);
        q1 += (b[0] + b[1]) + (b[2] + b[3]);
    }
    for (; s < S; ++s) {
        q0 += pt[(int64_t)s * (BK_ROWS * 2) + r];
        q1 += pt[(int64_t)s * (BK_ROWS * 2) + BK_ROWS + r];
    }
    const FixScale fs = fix_scale<FB>(gmax[l * CUR_STRIDE]);
    float g0 = ((float)q0 * fs.un_a) * fs.un_b, g1 = ((float)q1 * fs.un_a) * fs.un_b;
    const int64_t R = (int64_t)meta.offsets[l] + row0 + r;
    float2 *d2 = reinterpret_cast<float2 *>(dtable) + R;
    if (FUSE) {
        if (n_raw > cap) {  // overflow records of pass 1 sit in dtable: consume them, leave it zero
            const float2 d = *d2;
            g0 = d.x + g0;
            g1 = d.y + g1;
            *d2 = make_float2(0.f, 0.f);
        }
        if (fu.grad_out) {
            reinterpret_cast<uint32_t *>(fu.grad_out)[R] = (uint32_t)f32_to_bf16(g0) | ((uint32_t)f32_to_bf16(g1) << 16);
            return;
        }
        AdamArgs a = fu.a;
        adam_bias(a);
        a.zero_grad = 0;
        float2 P = reinterpret_cast<float2 *>(fu.p)[R], Mv = reinterpret_cast<float2 *>(fu.m)[R];
        float2 V = reinterpret_cast<float2 *>(fu.v)[R];
        adam_one(P.x, g0, Mv.x, V.x, a);
        adam_one(P.y, g1, Mv.y, V.y, a);
        reinterpret_cast<float2 *>(fu.p)[R] = P;
        reinterpret_cast<float2 *>(fu.m)[R] = Mv;
        reinterpret_cast<float2 *>(fu.v)[R] = V;
        if (fu.shadow) reinterpret_cast<uint32_t *>(fu.shadow)[R] = (uint32_t)f32_to_bf16(P.x) | ((uint32_t)f32_to_bf16(P.y) << 16);
    } else {
        float2 d = *d2;
        d.x += g0;
        d.y += g1;
        *d2 = d;
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

// device header of the workspace: bucket cursors (int32) followed by the per-level maxima (uint32), CUR_STRIDE apart
static size_t cursor_bytes(int n_buckets) {
    return ((size_t)(n_buckets + LNERF_MAX_LEVELS) * CUR_STRIDE * sizeof(int32_t) + 4095) / 4096 * 4096;
}

struct ScatterPlan {
    int64_t recs;       // record slots
    int buckets, wgs;   // buckets / pass-2 workgroups over all levels
    int ptiles;         // partial-sum tiles (sliced levels: buckets x slices)
    int fbuckets;       // buckets of sliced levels (grid of the finishing pass)
    size_t cursor_bytes, rec_bytes, partial_bytes;
    size_t total() const { return cursor_bytes + rec_bytes + partial_bytes; }
};

static int fill_bucket_meta(const GridMeta &meta, int64_t m_host, BucketMeta &bm, ScatterPlan &plan) {
    int64_t total_recs = 0;
    int total_buckets = 0, total_wgs = 0, total_ptiles = 0, total_fb = 0;
    for (int l = 0; l < meta.num_levels; ++l) {
        const int64_t hsize = meta.offsets[l + 1] - meta.offsets[l];
        const int nb = (int)div_up(hsize, BK_ROWS);
        if (nb > BK_MAX_PER_LEVEL) return -1;
        const int64_t per_bucket = div_up(8 * m_host, nb);
        // head-room over a uniform spread of the worst-case record count: 25 % on hashed levels (rows are spread
        // by the hash), 100 % on levels with few buckets (dense levels: a bucket is a slab of z layers, a compact
        // object loads the central slabs, and the last bucket of a level is only partly filled)
        int64_t cap = per_bucket + (nb <= 64 ? per_bucket : per_bucket / 4) + 1024;
        if (cap > 8 * m_host) cap = 8 * m_host;
        if (cap < 64) cap = 64;
        if (cap > 0x7FFFFFFF) return -1;
        if ((int64_t)nb * cap * (int64_t)sizeof(Rec12) >= (1ll << 32)) return -1;  // (the copy-out uses 32-bit byte offsets)
        int slices = (int)((per_bucket + 65535) / 65536);    // <= ~64 Ki records per pass-2 workgroup
        if (slices < 1) slices = 1;
        if (slices > 64) slices = 64;
        bm.nb[l] = nb;
        bm.bstart[l] = total_buckets;
        bm.cap[l] = (int)cap;
        bm.slices[l] = slices;
        bm.compact[l] = meta.res[l] <= g_compact_max_res ? 1 : 0;
        bm.wgstart[l] = total_wgs;
        bm.rstart[l] = total_recs;
        bm.pstart[l] = slices > 1 ? total_ptiles : -1;
        bm.fstart[l] = slices > 1 ? total_fb : -1;
        if (slices > 1) {
            total_ptiles += nb * slices;
            total_fb += nb;
        }
        total_buckets += nb;
        total_wgs += nb * slices;
        total_recs += (int64_t)nb * cap;
    }
    bm.bstart[meta.num_levels] = total_buckets;
    bm.wgstart[meta.num_levels] = total_wgs;
    plan.recs = total_recs;
    plan.buckets = total_buckets;
    plan.wgs = total_wgs;
    plan.ptiles = total_ptiles;
    plan.fbuckets = total_fb;
    plan.cursor_bytes = cursor_bytes(total_buckets);
    plan.rec_bytes = ((size_t)total_recs * sizeof(Rec12) + 15) / 16 * 16;  // (packed records use two thirds of it)
    plan.partial_bytes = (size_t)total_ptiles * BK_ROWS * 2 * sizeof(long long);
    return 0;
}

static int fill_meta(const char *who, GridMeta &meta, int num_levels, int level_dim, const int32_t *offsets_host,
                     const float *scales_host, const int32_t *res_host) {
    LNERF_REQUIRE(num_levels >= 1 && num_levels <= LNERF_MAX_LEVELS, "%s: num_levels out of range (%d)", who,
                  num_levels);
    LNERF_REQUIRE(level_dim == 2, "%s: only level_dim == 2 is built (got %d)", who, level_dim);
    LNERF_REQUIRE(offsets_host && scales_host && res_host, "%s: null level metadata", who);
    meta.num_levels = num_levels;
    for (int l = 0; l <= num_levels; ++l) meta.offsets[l] = offsets_host[l];
    for (int l = 0; l < num_levels; ++l) {
        LNERF_REQUIRE(offsets_host[l + 1] > offsets_host[l], "%s: empty level %d", who, l);
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
    int rc = fill_meta("grid_encode_forward", meta, num_levels, level_dim, offsets_host, scales_host, res_host);
    if (rc) return rc;
    LNERF_REQUIRE(m_host >= 0 && level_stride >= m_host, "grid_encode_forward: need 0 <= m_host <= level_stride");
    LNERF_REQUIRE(bound > 0.f, "grid_encode_forward: bound must be > 0");
    LNERF_REQUIRE(variant >= 0 && variant <= 2, "grid_encode_forward: unknown variant %d", variant);
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
        g_mlp_fwd_wps = value;
        return LNERF_OK;
    }
    if (strcmp(key, "mlp_fwd_blocks") == 0) {
        LNERF_REQUIRE(value >= 1 && value <= 65535, "set_tuning: mlp_fwd_blocks out of range");
        g_mlp_fwd_blocks = value;
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
    for (int l = 0; l <= num_levels; ++l) meta.offsets[l] = offsets_host[l];
    for (int l = 0; l < num_levels; ++l) meta.res[l] = 0;
    BucketMeta bm;
    ScatterPlan plan;
    if (fill_bucket_meta(meta, m_host, bm, plan) != 0) return 0;
    return plan.cursor_bytes;
}

size_t lnerf_grid_encode_backward_workspace_bytes(int num_levels, const int32_t *offsets_host, int64_t m_host) {
    if (num_levels < 1 || num_levels > LNERF_MAX_LEVELS || !offsets_host || m_host < 0) return 0;
    GridMeta meta;
    meta.num_levels = num_levels;
    for (int l = 0; l <= num_levels; ++l) meta.offsets[l] = offsets_host[l];
    for (int l = 0; l < num_levels; ++l) meta.res[l] = 0;
    BucketMeta bm;
    ScatterPlan plan;
    if (fill_bucket_meta(meta, m_host, bm, plan) != 0) return 0;
    return plan.total();  // bucket cursors, the records, the partial-sum tiles of the sliced levels
}

// fu == nullptr: dtable += scatter.  fu != nullptr: every row's Adam step is applied by whichever kernel finishes its
// sum (pass 2 on unsliced levels, the finishing pass on sliced ones); dtable only carries overflow records.
// phases: 1 = pass 1 (binning, all levels; clears the cursors), 2 = pass 2 + finishing pass of levels [lv_lo, lv_hi)
static int scatter_backward(const float *xyzs, float bound, const void *dfeat, int dfeat_dtype, int num_levels,
                            int level_dim, const int32_t *offsets_host, const float *scales_host,
                            const int32_t *res_host, int64_t m_host, const int32_t *m_dev, int64_t level_stride,
                            float *dtable, int variant, void *workspace, size_t workspace_bytes,
                            lnerf_stream_t stream, FusedUpdate *fu, int phases = 3, int lv_lo = 0, int lv_hi = -1) {
    // LNERF_SCATTER_CLEARED: the caller zeroed the head of the workspace (lnerf_grid_scatter_clear_bytes()) with
    // something it was launching anyway -- the fill dispatch of this call is skipped
    const bool cleared = (variant & LNERF_SCATTER_CLEARED) != 0;
    variant &= ~LNERF_SCATTER_CLEARED;
    GridMeta meta;
    int rc = fill_meta("grid_encode_backward", meta, num_levels, level_dim, offsets_host, scales_host, res_host);
    if (rc) return rc;
    LNERF_REQUIRE(m_host >= 0 && level_stride >= m_host, "grid_encode_backward: need 0 <= m_host <= level_stride");
    LNERF_REQUIRE(m_host < (1ll << 30), "grid_encode_backward: m_host must be below 2^30 samples");
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
    const int nbk = plan.buckets;
    const size_t cbytes = plan.cursor_bytes;
    const size_t need = plan.total();
    LNERF_REQUIRE(workspace && workspace_bytes >= need, "grid_encode_backward: workspace too small (%zu < %zu)",
                  workspace_bytes, need);
    LNERF_REQUIRE(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)dtable & 15) == 0,
                  "grid_encode_backward: workspace/dtable must be 16-byte aligned");
    int32_t *cursor = (int32_t *)workspace;
    unsigned int *gmax = (unsigned int *)workspace + (size_t)nbk * CUR_STRIDE;
    void *rec = (char *)workspace + cbytes;
    long long *partials = (long long *)((char *)workspace + cbytes + plan.rec_bytes);
    const bool packed = variant == 3;
    if ((phases & 1) && !cleared && hipMemsetAsync(cursor, 0, cbytes, s) != hipSuccess) {
        set_error("grid_encode_backward: hipMemsetAsync failed");
        return LNERF_ERR_HIP;
    }
    auto launch_bin = [&]() {
        // persistent: G workgroups, G a multiple of the level count (item k of a workgroup: next tile group, next level)
        const int wgs = g_bin_wgs > 0 ? g_bin_wgs : 256 * g_bin_per_cu;
        int64_t G = (int64_t)(wgs / num_levels) * num_levels;
        const int64_t items = div_up(m_host, (int64_t)BIN_T) * num_levels;
        if (G > items) G = items;
        if (G < num_levels) G = num_levels;
        const dim3 g((unsigned)G, 1, 1);
        if (packed)
            hipLaunchKernelGGL((k_scatter_bin<Rec8>), g, dim3(BIN_T), 0, s, xyzs, bound, (const float *)dfeat, meta, bm,
                               m_host, m_dev, level_stride, cursor, gmax, (Rec8 *)rec, dtable, g_skip_zero);
        else
            hipLaunchKernelGGL((k_scatter_bin<Rec12>), g, dim3(BIN_T), 0, s, xyzs, bound, (const float *)dfeat, meta, bm,
                               m_host, m_dev, level_stride, cursor, gmax, (Rec12 *)rec, dtable, g_skip_zero);
    };
    FusedUpdate fu0;
    memset(&fu0, 0, sizeof(fu0));
    if (fu) fu0 = *fu;
    auto launch_reduce = [&](hipStream_t st, int l0, int l1) {
        const int w0 = bm.wgstart[l0], w1 = bm.wgstart[l1];
        if (w1 <= w0) return;
#define LAUNCH_RED(T, REC, FUSE)                                                                                  \
    hipLaunchKernelGGL((k_scatter_reduce<T, REC, FUSE>), dim3((unsigned)(w1 - w0)), dim3(T), 0, st, meta, bm, cursor, \
                       gmax, (const REC *)rec, dtable, partials, w0, fu0)
        if (fu && packed) LAUNCH_RED(1024, Rec8, true);
        else if (fu) LAUNCH_RED(1024, Rec12, true);
        else if (packed && g_reduce_threads == 512) LAUNCH_RED(512, Rec8, false);
        else if (packed) LAUNCH_RED(1024, Rec8, false);
        else if (g_reduce_threads == 512) LAUNCH_RED(512, Rec12, false);
        else LAUNCH_RED(1024, Rec12, false);
#undef LAUNCH_RED
    };
    auto launch_finish = [&]() {  // sliced levels: add up the slices' exact partial sums
        if (plan.fbuckets == 0) return;
        const dim3 g((unsigned)(plan.fbuckets * (BK_ROWS / 256)));
#define LAUNCH_FIN(FUSE, FB)                                                                                        \
    hipLaunchKernelGGL((k_scatter_finish<FUSE, FB>), g, dim3(256), 0, s, meta, bm, cursor, gmax, partials, dtable, fu0,   \
                       lv_lo, lv_hi)
        if (fu && packed) LAUNCH_FIN(true, 30);
        else if (fu) LAUNCH_FIN(true, 44);
        else if (packed) LAUNCH_FIN(false, 30);
        else LAUNCH_FIN(false, 44);
#undef LAUNCH_FIN
    };
    if (phases & 1) {
        launch_bin();
        LNERF_CHECK_LAUNCH("grid_encode_backward(bin)");
    }
    if (phases & 2) {
        launch_reduce(s, lv_lo, lv_hi);
        LNERF_CHECK_LAUNCH("grid_encode_backward(reduce)");
        bool any_sliced = false;
        for (int l = lv_lo; l < lv_hi; ++l) any_sliced = any_sliced || bm.slices[l] > 1;
        if (any_sliced) launch_finish();
        LNERF_CHECK_LAUNCH("grid_encode_backward(finish)");
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

}  // extern "C"
