// H6, pass 2 of the bucketed scatter (fixed-point LDS sums, fused Adam step of the table, the step's tail), the host-side
// driver of both passes and the C entry points of the scatter.  Gather: grid_gather.hip; pass 1: grid_bin.hip; what
// the passes share: grid_shared.h.
#include "grid_shared.h"

namespace lnerf {

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
int g_compact_max_res = 512;
// gather: fetch x-adjacent vertices with one load where they are adjacent rows (2: also aligned groups of four rows)
int g_gather_pairs = 2;
// gather variant 2: workgroups per XCD (each strides over the tiles of its XCD's levels)
int g_gather_wgs_per_xcd = 256;
// gather: levels with resolution <= this fetch a cell's vertices once per run of lanes in that cell (0 = off)
int g_gather_dedup_res = 512;
// gather: bytes of (unused) dynamic LDS per workgroup -- an EXPERIMENT knob that caps the resident wavefronts (160 KiB per CU:
// 53 KiB leaves three 256-thread workgroups = three waves per SIMD): "what would the gather cost at the occupancy of a
// kernel fused with the MLP forward?" 
int g_gather_lds_pad = 0;
// persistent workgroups of the binning pass per CU (3 fit its 44 KiB of LDS with the 8-byte records)
int g_bin_per_cu = 3;
// persistent workgroups of the binning pass (0 = 256 CUs x g_bin_per_cu); rounded down to a multiple of the level count
int g_bin_wgs = 0;
// drop contributions that are exactly zero (samples behind a ray's termination point)
int g_skip_zero = 1;
// threads per workgroup of the reduce pass (512 or 1024; two 64 KiB workgroups fit a CU either way)
int g_reduce_threads = 1024;
// level groups of the whole-frame scatter: bin(group) -> reduce(group) per group (1 = bin everything, then reduce)
int g_scatter_groups = 1;

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

}  // namespace lnerf

using namespace lnerf;

extern "C" {

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
    if (strcmp(key, "gather_lds_pad") == 0) {
        LNERF_REQUIRE(value >= 0 && value <= 65536, "set_tuning: gather_lds_pad must be in [0, 65536] bytes");
        g_gather_lds_pad = value;
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
int lnerf_debug_bin_stamps(unsigned long long *out16) {   // pass 1's totals (grid_bin.hip) + pass 2's (this file); clears both
    unsigned long long z[16] = {0}, mine[16];
    if (hipMemcpyFromSymbol(mine, HIP_SYMBOL(g_bin_stamps), sizeof(z)) != hipSuccess) return LNERF_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_bin_stamps), z, sizeof(z)) != hipSuccess) return LNERF_ERR_HIP;
    const int rc = bin_stamps_read(out16);
    if (rc != LNERF_OK) return rc;
    for (int k = 0; k < 16; ++k) out16[k] += mine[k];
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
    if (variant < 2) {
        launch_grid_backward_atomic(xyzs, bound, (const float *)dfeat, meta, m_host, m_dev, level_stride, dtable, variant, s);
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
    auto launch_bin = [&](int l0, int l1) {   // pass 1 (grid_bin.hip)
        launch_scatter_bin(packed, xyzs, bound, (const float *)dfeat, meta, bm, m_host, m_dev, level_stride, gmax, items_dev,
                           segtab, rec, l0, l1, s);
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
