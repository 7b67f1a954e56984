// H6, pass 1 of the bucketed scatter (see grid_shared.h for the layout both passes share).
#include "grid_shared.h"

namespace lnerf {

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


void launch_scatter_bin(bool packed, const float *xyzs, float bound, const float *dfeat, const GridMeta &meta,
                        const BucketMeta &bm, int64_t m_host, const int32_t *m_dev, int64_t level_stride, unsigned int *gmax,
                        int32_t *items_dev, uint32_t *segtab, void *rec, int l0, int l1, hipStream_t s) {
    // persistent: G workgroups, G a multiple of the launch's level count (item k of a workgroup: next tile group, next level)
    const int nl = l1 - l0;
    const int wgs = g_bin_wgs > 0 ? g_bin_wgs : 256 * g_bin_per_cu;
    int64_t G = (int64_t)(wgs / nl) * nl;
    const int64_t items = div_up(m_host, (int64_t)BIN_T) * nl;
    if (G > items) G = items;
    if (G < nl) G = nl;
    const dim3 g((unsigned)G, 1, 1);
    if (packed)
        hipLaunchKernelGGL((k_scatter_bin<Rec8>), g, dim3(BIN_T), 0, s, xyzs, bound, dfeat, meta, bm, m_host, m_dev,
                           level_stride, gmax, items_dev, segtab, (Rec8 *)rec, g_skip_zero, l0, l1);
    else
        hipLaunchKernelGGL((k_scatter_bin<Rec12>), g, dim3(BIN_T), 0, s, xyzs, bound, dfeat, meta, bm, m_host, m_dev,
                           level_stride, gmax, items_dev, segtab, (Rec12 *)rec, g_skip_zero, l0, l1);
}

#ifdef LNERF_STAMPS
int bin_stamps_read(unsigned long long *out16) {
    unsigned long long z[16] = {0};
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bin_stamps), sizeof(z)) != hipSuccess) return LNERF_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bin_stamps), z, sizeof(z)) != hipSuccess) return LNERF_ERR_HIP;
    return LNERF_OK;
}
#endif

}  // namespace lnerf
