// Shared definitions of the multiresolution hash grid's translation units (H5 / H6):
//   grid_gather.hip   forward gather (the roofline kernel) + the atomic reference scatter
//   grid_bin.hip      pass 1 of the bucketed scatter (records binned per bucket, item chunks)
//   grid.hip          pass 2 (fixed-point sums, fused Adam, the step's tail), the host-side driver and the C entry points
#pragma once
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
        const uint32_t y0 = (gy >> 1) * 2654435761u, z0 = (gz >> 1) * 805459861u;
        // (y + 1) >> 1 is the next block exactly when y is odd
        const uint32_t y1 = (gy & 1u) ? y0 + 2654435761u : y0, z1 = (gz & 1u) ? z0 + 805459861u : z0;
        if ((nblk & (nblk - 1u)) == 0u) {
            // The position inside the block rides in the low four bits of the three hash terms: x in bits 0-1, y in bit
            // 2, z in bit 3 -- disjoint, so their XOR is their OR -- and one mask keeps (block hash mod nblk) << 4 and
            // the position: a corner costs one three-input XOR and one AND, as on the vertex-hash path.  (The binning
            // pass is bound by vector issue: the shift / or / or per corner of the plain form cost it 7 us.)
            const uint32_t HX[2] = {((gx & ~3u) << 2) | (gx & 3u), ((x1 & ~3u) << 2) | (x1 & 3u)};
            const uint32_t HY[2] = {(y0 << 4) | ((gy & 1u) << 2), (y1 << 4) | (((gy + 1u) & 1u) << 2)};
            const uint32_t HZ[2] = {(z0 << 4) | ((gz & 1u) << 3), (z1 << 4) | (((gz + 1u) & 1u) << 3)};
            const uint32_t mask = ((nblk - 1u) << 4) | 15u;
#pragma unroll
            for (int c = 0; c < 8; ++c) row[c] = (HX[c & 1] ^ HY[(c >> 1) & 1] ^ HZ[(c >> 2) & 1]) & mask;
        } else {
            const uint32_t hx[2] = {gx >> 2, x1 >> 2};
            const uint32_t hy[2] = {y0, y1};
            const uint32_t hz[2] = {z0, z1};
            const uint32_t wx[2] = {gx & 3u, x1 & 3u};
            const uint32_t wy[2] = {(gy & 1u) << 2, ((gy + 1u) & 1u) << 2};
            const uint32_t wz[2] = {(gz & 1u) << 3, ((gz + 1u) & 1u) << 3};
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
static __device__ unsigned long long g_bin_stamps[16];   // (one copy per translation unit: bin 0..9, reduce 10..15)
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
static __device__ unsigned long long g_wg_log[4 * 4096];
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


// ---- tunables (lnerf_set_tuning; defined in grid.hip)
extern int g_compact_max_res, g_gather_pairs, g_gather_wgs_per_xcd, g_gather_dedup_res, g_bin_per_cu, g_bin_wgs, g_skip_zero,
    g_reduce_threads, g_scatter_groups, g_gather_lds_pad;

static inline int fill_meta(const char *who, GridMeta &meta, int num_levels, int level_dim, const int32_t *offsets_host,
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


// ---- launchers of the other translation units
void launch_grid_backward_atomic(const float *xyzs, float bound, const float *dfeat, const GridMeta &meta, int64_t m_host,
                                 const int32_t *m_dev, int64_t level_stride, float *dtable, int variant, hipStream_t s);
void launch_scatter_bin(bool packed, const float *xyzs, float bound, const float *dfeat, const GridMeta &meta,
                        const BucketMeta &bm, int64_t m_host, const int32_t *m_dev, int64_t level_stride, unsigned int *gmax,
                        int32_t *items_dev, uint32_t *segtab, void *rec, int l0, int l1, hipStream_t s);
#ifdef LNERF_STAMPS
int bin_stamps_read(unsigned long long *out16);   // (grid_bin.hip: its copy of the stamp totals, cleared by the read)
#endif

}  // namespace lnerf
