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
#include "grid_shared.h"

namespace lnerf {

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

void launch_grid_backward_atomic(const float *xyzs, float bound, const float *dfeat, const GridMeta &meta, int64_t m_host,
                                 const int32_t *m_dev, int64_t level_stride, float *dtable, int variant, hipStream_t s) {
    dim3 grid;
    launch_dims(variant, meta.num_levels, m_host, grid);
    hipLaunchKernelGGL((k_grid_backward_atomic<float>), grid, dim3(256), 0, s, xyzs, bound, dfeat, meta, m_host, m_dev,
                       level_stride, dtable, variant);
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
    hipLaunchKernelGGL((k_grid_forward<TT, TO>), grid, dim3(256), (size_t)g_gather_lds_pad, s, xyzs, bound, (const TT *)table, meta, m_host, \
                       m_dev, level_stride, (TO *)feat, variant, g_gather_pairs, g_gather_dedup_res, plan)
    if (table_dtype == LNERF_F32 && feat_dtype == LNERF_F32) LAUNCH_FWD(float, float);
    else if (table_dtype == LNERF_F32) LAUNCH_FWD(float, uint16_t);
    else if (feat_dtype == LNERF_F32) LAUNCH_FWD(uint16_t, float);
    else LAUNCH_FWD(uint16_t, uint16_t);
#undef LAUNCH_FWD
    LNERF_CHECK_LAUNCH("grid_encode_forward");
    return LNERF_OK;
}

}  // extern "C"
