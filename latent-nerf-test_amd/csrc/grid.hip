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

__device__ __forceinline__ LevelPos level_pos(const float *__restrict__ xyzs, int64_t m, float bound, float scale) {
    // x01 = (x + bound) / (2 bound); pos = x01 * scale + 0.5   (op order = oracle grid_encode)
    const float two_b = 2.0f * bound;
    float px = (xyzs[m * 3] + bound) / two_b;
    float py = (xyzs[m * 3 + 1] + bound) / two_b;
    float pz = (xyzs[m * 3 + 2] + bound) / two_b;
    px = px * scale; py = py * scale; pz = pz * scale;
    px = px + 0.5f; py = py + 0.5f; pz = pz + 0.5f;
    const float flx = floorf(px), fly = floorf(py), flz = floorf(pz);
    LevelPos r;
    r.gx = (uint32_t)(int)flx; r.gy = (uint32_t)(int)fly; r.gz = (uint32_t)(int)flz;
    r.fx = px - flx; r.fy = py - fly; r.fz = pz - flz;
    return r;
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

template <typename TT, typename TO>
__global__ void __launch_bounds__(256)
k_grid_forward(const float *__restrict__ xyzs, float bound, const TT *__restrict__ table, GridMeta meta, int64_t m_host,
               const int32_t *__restrict__ m_dev, int64_t level_stride, TO *__restrict__ feat, int variant) {
    int64_t M = m_host;
    if (m_dev) { const int64_t md = *m_dev; M = md < M ? md : M; }
    const TileMap tm = tile_map(variant, meta.num_levels);
    if (!tm.ok) return;
    const int l = tm.level;
    const float scale = meta.scales[l];
    const uint32_t res = (uint32_t)meta.res[l];
    const uint32_t off = (uint32_t)meta.offsets[l];
    const uint32_t hsize = (uint32_t)(meta.offsets[l + 1] - meta.offsets[l]);
    const TT *lt = table + (int64_t)off * 2;
    for (int64_t tile = tm.tile0; tile * 256 < M; tile += tm.tstep) {
        const int64_t m = tile * 256 + threadIdx.x;
        if (m >= M) continue;
        const LevelPos p = level_pos(xyzs, m, bound, scale);
        // issue the 8 gathers first, blend afterwards (keeps 8 loads in flight per lane)
        float2 v[8];
        float w[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t bx = c & 1, by = (c >> 1) & 1, bz = (c >> 2) & 1;
            const uint32_t row = grid_index(p.gx + bx, p.gy + by, p.gz + bz, res, hsize);
            v[c] = Feat2<TT>::load(lt, row);
            const float wx = bx ? p.fx : 1.0f - p.fx;
            const float wy = by ? p.fy : 1.0f - p.fy;
            const float wz = bz ? p.fz : 1.0f - p.fz;
            w[c] = (wx * wy) * wz;
        }
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            a0 = fmaf(w[c], v[c].x, a0);
            a1 = fmaf(w[c], v[c].y, a1);
        }
        Feat2<TO>::store(feat, (int64_t)l * level_stride + m, a0, a1);
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
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint32_t bx = c & 1, by = (c >> 1) & 1, bz = (c >> 2) & 1;
            const uint32_t row = grid_index(p.gx + bx, p.gy + by, p.gz + bz, res, hsize);
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
//   pass 1 (k_scatter_bin)    one thread per (sample, level) computes its 8 (row, w*g) records,
//                             ranks them inside the workgroup tile with LDS counters, reserves a
//                             span per touched bucket with ONE returning global atomic, and
//                             appends the 16-byte records to the bucket's region with plain stores;
//   pass 2 (k_scatter_reduce) one workgroup per (bucket, slice) streams its records (coalesced
//                             16 B/lane), accumulates them with LDS float atomics and adds the
//                             64 KiB tile to the gradient table with coalesced stores.
// A bucket that overflows its region falls back to global atomics for the excess records, so the
// result is always complete.
constexpr int BK_SHIFT = 13, BK_ROWS = 1 << BK_SHIFT;  // 8192 rows * 8 B = 64 KiB
constexpr int BK_MAX_PER_LEVEL = 256;                   // LDS counters per workgroup tile

struct BucketMeta {
    int nb[LNERF_MAX_LEVELS];            // buckets per level
    int bstart[LNERF_MAX_LEVELS + 1];    // first global bucket id of the level
    int cap[LNERF_MAX_LEVELS];           // record capacity of each bucket of the level
    int slices[LNERF_MAX_LEVELS];        // pass-2 workgroups per bucket
    int wgstart[LNERF_MAX_LEVELS + 1];   // first pass-2 workgroup of the level
    long long rstart[LNERF_MAX_LEVELS];  // first record slot of the level's region
};

template <typename TG>
__global__ void __launch_bounds__(256)
k_scatter_bin(const float *__restrict__ xyzs, float bound, const TG *__restrict__ dfeat, GridMeta meta, BucketMeta bm,
              int64_t m_host, const int32_t *__restrict__ m_dev, int64_t level_stride, int32_t *__restrict__ cursor,
              uint4 *__restrict__ recs, float *__restrict__ dtable, int variant) {
    __shared__ int s_cnt[BK_MAX_PER_LEVEL];
    __shared__ int s_base[BK_MAX_PER_LEVEL];
    int64_t M = m_host;
    if (m_dev) { const int64_t md = *m_dev; M = md < M ? md : M; }
    const TileMap tm = tile_map(variant, meta.num_levels);
    if (!tm.ok) return;
    const int l = tm.level;
    const float scale = meta.scales[l];
    const uint32_t res = (uint32_t)meta.res[l];
    const uint32_t off = (uint32_t)meta.offsets[l];
    const uint32_t hsize = (uint32_t)(meta.offsets[l + 1] - meta.offsets[l]);
    const int nb = bm.nb[l], cap = bm.cap[l], b0 = bm.bstart[l];
    uint4 *lrec = recs + bm.rstart[l];
    float *lt = dtable + (int64_t)off * 2;
    const int tid = threadIdx.x;
    for (int64_t tile = tm.tile0; tile * 256 < M; tile += tm.tstep) {
        for (int i = tid; i < nb; i += 256) s_cnt[i] = 0;
        __syncthreads();
        const int64_t m = tile * 256 + tid;
        const bool valid = m < M;
        uint32_t row[8];
        float wv[8];
        int rank[8];
        float2 gg = make_float2(0.f, 0.f);
        if (valid) {
            const LevelPos p = level_pos(xyzs, m, bound, scale);
            gg = Feat2<TG>::load(dfeat + ((int64_t)l * level_stride + m) * 2, 0);
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const uint32_t bx = c & 1, by = (c >> 1) & 1, bz = (c >> 2) & 1;
                row[c] = grid_index(p.gx + bx, p.gy + by, p.gz + bz, res, hsize);
                const float wx = bx ? p.fx : 1.0f - p.fx;
                const float wy = by ? p.fy : 1.0f - p.fy;
                const float wz = bz ? p.fz : 1.0f - p.fz;
                wv[c] = (wx * wy) * wz;
                rank[c] = atomicAdd(&s_cnt[row[c] >> BK_SHIFT], 1);
            }
        }
        __syncthreads();
        for (int i = tid; i < nb; i += 256) {
            const int c = s_cnt[i];
            s_base[i] = c ? atomicAdd(&cursor[b0 + i], c) : 0;
        }
        __syncthreads();
        if (valid) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int b = (int)(row[c] >> BK_SHIFT);
                const int slot = s_base[b] + rank[c];
                const float v0 = wv[c] * gg.x, v1 = wv[c] * gg.y;
                if (slot < cap) {
                    lrec[(int64_t)b * cap + slot] =
                        make_uint4(row[c] & (BK_ROWS - 1), __float_as_uint(v0), __float_as_uint(v1), 0u);
                } else {  // bucket region full: finish this record with global atomics
                    atomicAdd(lt + (int64_t)row[c] * 2, v0);
                    atomicAdd(lt + (int64_t)row[c] * 2 + 1, v1);
                }
            }
        }
        // s_cnt is re-zeroed behind the next iteration's first barrier; s_base readers are done
        // before anyone can pass that barrier and the one after it.
    }
}

__global__ void __launch_bounds__(512)
k_scatter_reduce(GridMeta meta, BucketMeta bm, const int32_t *__restrict__ cursor, const uint4 *__restrict__ recs,
                 float *__restrict__ dtable) {
    __shared__ float acc[BK_ROWS * 2];
    // locate (level, bucket, slice) of this workgroup
    int l = 0;
    while (l + 1 < meta.num_levels && (int)blockIdx.x >= bm.wgstart[l + 1]) ++l;
    const int S = bm.slices[l];
    const int local = (int)blockIdx.x - bm.wgstart[l];
    const int b = local / S, s = local - b * S;
    const int cap = bm.cap[l];
    int n = cursor[bm.bstart[l] + b];
    n = n < cap ? n : cap;
    const int lo = (int)(((long long)n * s) / S), hi = (int)(((long long)n * (s + 1)) / S);
    if (hi <= lo) return;  // uniform per workgroup
    const int tid = threadIdx.x;
    for (int i = tid; i < BK_ROWS * 2; i += 512) acc[i] = 0.f;
    __syncthreads();
    const uint4 *rp = recs + bm.rstart[l] + (long long)b * cap;
    for (int i = lo + tid; i < hi; i += 512) {
        const uint4 r = rp[i];
        atomicAdd(&acc[r.x * 2], __uint_as_float(r.y));
        atomicAdd(&acc[r.x * 2 + 1], __uint_as_float(r.z));
    }
    __syncthreads();
    const int hsize = meta.offsets[l + 1] - meta.offsets[l];
    const int row0 = b << BK_SHIFT;
    int rows = hsize - row0;
    rows = rows < BK_ROWS ? rows : BK_ROWS;
    float *dst = dtable + ((int64_t)meta.offsets[l] + row0) * 2;
    if (S == 1) {  // sole owner of these rows in this launch: plain read-modify-write, 16 B per lane
        const int n4 = (rows * 2) >> 2;
        for (int i = tid; i < n4; i += 512) {
            float4 d = reinterpret_cast<float4 *>(dst)[i];
            const float4 a = reinterpret_cast<const float4 *>(acc)[i];
            d.x += a.x; d.y += a.y; d.z += a.z; d.w += a.w;
            reinterpret_cast<float4 *>(dst)[i] = d;
        }
        for (int i = (n4 << 2) + tid; i < rows * 2; i += 512) dst[i] += acc[i];
    } else {       // several slices share the rows: contiguous float atomics (256 B per wave instruction)
        for (int i = tid; i < rows * 2; i += 512) {
            const float a = acc[i];
            if (a != 0.f) atomicAdd(&dst[i], a);
        }
    }
}

static int fill_bucket_meta(const GridMeta &meta, int64_t m_host, BucketMeta &bm, int64_t &total_recs,
                            int &total_buckets, int &total_wgs) {
    total_recs = 0;
    total_buckets = 0;
    total_wgs = 0;
    for (int l = 0; l < meta.num_levels; ++l) {
        const int64_t hsize = meta.offsets[l + 1] - meta.offsets[l];
        const int nb = (int)div_up(hsize, BK_ROWS);
        if (nb > BK_MAX_PER_LEVEL) return -1;
        const int64_t per_bucket = div_up(8 * m_host, nb);
        int64_t cap = per_bucket + per_bucket / 4 + 1024;  // 25 % head-room over a uniform spread
        if (cap > 8 * m_host) cap = 8 * m_host;
        if (cap < 64) cap = 64;
        if (cap > 0x7FFFFFFF) return -1;
        int slices = (int)((per_bucket + 65535) / 65536);    // ~64 Ki records per pass-2 workgroup
        if (slices < 1) slices = 1;
        if (slices > 64) slices = 64;
        bm.nb[l] = nb;
        bm.bstart[l] = total_buckets;
        bm.cap[l] = (int)cap;
        bm.slices[l] = slices;
        bm.wgstart[l] = total_wgs;
        bm.rstart[l] = total_recs;
        total_buckets += nb;
        total_wgs += nb * slices;
        total_recs += (int64_t)nb * cap;
    }
    bm.bstart[meta.num_levels] = total_buckets;
    bm.wgstart[meta.num_levels] = total_wgs;
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
    LNERF_REQUIRE(variant == 0 || variant == 1, "grid_encode_forward: unknown variant %d", variant);
    LNERF_REQUIRE((table_dtype == LNERF_F32 || table_dtype == LNERF_BF16) &&
                      (feat_dtype == LNERF_F32 || feat_dtype == LNERF_BF16),
                  "grid_encode_forward: bad dtype tag");
    if (m_host == 0) return LNERF_OK;
    LNERF_REQUIRE(xyzs && table && feat, "grid_encode_forward: null pointer");
    dim3 grid;
    launch_dims(variant, num_levels, m_host, grid);
    hipStream_t s = as_stream(stream);
#define LAUNCH_FWD(TT, TO)                                                                                         \
    hipLaunchKernelGGL((k_grid_forward<TT, TO>), grid, dim3(256), 0, s, xyzs, bound, (const TT *)table, meta, m_host, \
                       m_dev, level_stride, (TO *)feat, variant)
    if (table_dtype == LNERF_F32 && feat_dtype == LNERF_F32) LAUNCH_FWD(float, float);
    else if (table_dtype == LNERF_F32) LAUNCH_FWD(float, uint16_t);
    else if (feat_dtype == LNERF_F32) LAUNCH_FWD(uint16_t, float);
    else LAUNCH_FWD(uint16_t, uint16_t);
#undef LAUNCH_FWD
    LNERF_CHECK_LAUNCH("grid_encode_forward");
    return LNERF_OK;
}

size_t lnerf_grid_encode_backward_workspace_bytes(int num_levels, const int32_t *offsets_host, int64_t m_host) {
    if (num_levels < 1 || num_levels > LNERF_MAX_LEVELS || !offsets_host || m_host < 0) return 0;
    GridMeta meta;
    meta.num_levels = num_levels;
    for (int l = 0; l <= num_levels; ++l) meta.offsets[l] = offsets_host[l];
    BucketMeta bm;
    int64_t recs;
    int nbk, nwg;
    if (fill_bucket_meta(meta, m_host, bm, recs, nbk, nwg) != 0) return 0;
    return (size_t)4096 + (size_t)recs * sizeof(uint4);  // [0,4096): bucket cursors, then the records
}

int lnerf_grid_encode_backward(const float *xyzs, float bound, const void *dfeat, int dfeat_dtype, int num_levels,
                               int level_dim, const int32_t *offsets_host, const float *scales_host,
                               const int32_t *res_host, int64_t m_host, const int32_t *m_dev, int64_t level_stride,
                               float *dtable, int variant, void *workspace, size_t workspace_bytes,
                               lnerf_stream_t stream) {
    GridMeta meta;
    int rc = fill_meta("grid_encode_backward", meta, num_levels, level_dim, offsets_host, scales_host, res_host);
    if (rc) return rc;
    LNERF_REQUIRE(m_host >= 0 && level_stride >= m_host, "grid_encode_backward: need 0 <= m_host <= level_stride");
    LNERF_REQUIRE(bound > 0.f, "grid_encode_backward: bound must be > 0");
    LNERF_REQUIRE(variant >= 0 && variant <= 2, "grid_encode_backward: unknown variant %d", variant);
    LNERF_REQUIRE(dfeat_dtype == LNERF_F32, "grid_encode_backward: dfeat must be f32");
    if (m_host == 0) return LNERF_OK;
    LNERF_REQUIRE(xyzs && dfeat && dtable, "grid_encode_backward: null pointer");
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
    int64_t recs;
    int nbk, nwg;
    LNERF_REQUIRE(fill_bucket_meta(meta, m_host, bm, recs, nbk, nwg) == 0,
                  "grid_encode_backward: level too large for the bucketed scatter (use variant 0/1)");
    LNERF_REQUIRE(nbk * (int)sizeof(int32_t) <= 4096, "grid_encode_backward: too many buckets (%d)", nbk);
    const size_t need = (size_t)4096 + (size_t)recs * sizeof(uint4);
    LNERF_REQUIRE(workspace && workspace_bytes >= need, "grid_encode_backward: workspace too small (%zu < %zu)",
                  workspace_bytes, need);
    LNERF_REQUIRE(((uintptr_t)workspace & 15) == 0 && ((uintptr_t)dtable & 15) == 0,
                  "grid_encode_backward: workspace/dtable must be 16-byte aligned");
    int32_t *cursor = (int32_t *)workspace;
    uint4 *rec = (uint4 *)((char *)workspace + 4096);
    if (hipMemsetAsync(cursor, 0, 4096, s) != hipSuccess) {
        set_error("grid_encode_backward: hipMemsetAsync failed");
        return LNERF_ERR_HIP;
    }
    launch_dims(1, num_levels, m_host, grid);
    hipLaunchKernelGGL((k_scatter_bin<float>), grid, dim3(256), 0, s, xyzs, bound, (const float *)dfeat, meta, bm,
                       m_host, m_dev, level_stride, cursor, rec, dtable, 1);
    LNERF_CHECK_LAUNCH("grid_encode_backward(bin)");
    hipLaunchKernelGGL(k_scatter_reduce, dim3((unsigned)nwg), dim3(512), 0, s, meta, bm, cursor, rec, dtable);
    LNERF_CHECK_LAUNCH("grid_encode_backward(reduce)");
    return LNERF_OK;
}

}  // extern "C"
