// H1 ray generation, H2 AABB slab test, H3 Morton/bitfield, H4 occupancy-pruned march
// (training: count -> scan -> write; inference: march_rays / compact_rays).
// gfx950 only: wave64 ballot / mbcnt compaction, one wavefront per ray.
#include "common.h"

#include <math.h>
#include <string.h>

namespace lnerf {

// ------------------------------------------------------------------ H1
// pixel g of a batch of B views -> ray (origin, unit direction): camera-to-world columns are right / down / forward / eye
__device__ __forceinline__ void pixel_ray(const float *__restrict__ c2w, int H, int W, float fx, float fy, float cx,
                                          float cy, int64_t g, float o[3], float d[3]) {
    const int b = (int)(g / ((int64_t)H * W));
    const int p = (int)(g - (int64_t)b * H * W);
    const int j = p / W, i = p - j * W;
    const float *m = c2w + (int64_t)b * 16;
    const float xs = ((float)i + 0.5f - cx) / fx;
    const float ys = ((float)j + 0.5f - cy) / fy;
    const float inv = 1.0f / sqrtf(xs * xs + ys * ys + 1.0f);
    const float d0 = xs * inv, d1 = ys * inv, d2 = inv;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        d[r] = d0 * m[r * 4 + 0] + d1 * m[r * 4 + 1] + d2 * m[r * 4 + 2];
        o[r] = m[r * 4 + 3];
    }
}
__global__ void __launch_bounds__(256) k_get_rays(const float *__restrict__ c2w, int B, int H, int W, float fx,
                                                  float fy, float cx, float cy, float *__restrict__ rays_o,
                                                  float *__restrict__ rays_d) {
    const int64_t total = (int64_t)B * H * W;
    for (int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; g < total; g += (int64_t)gridDim.x * blockDim.x) {
        float o[3], d[3];
        pixel_ray(c2w, H, W, fx, fy, cx, cy, g, o, d);
#pragma unroll
        for (int r = 0; r < 3; ++r) { rays_o[g * 3 + r] = o[r]; rays_d[g * 3 + r] = d[r]; }
    }
}
// ray generation folded into the march's count pass (lnerf_march_rays_train_pose): camera + the buffers the rays go to
struct RayGen {
    const float *c2w;
    const float *intr;   // optional device intrinsics [B,4] = (fx, fy, cx, cy) per view: override the by-value ones, so
                         // that a replayed hipGraph can render a new camera every time (lnerf_march_rays_train_camera)
    int on, H, W;
    float fx, fy, cx, cy;
    float *ro, *rd;
};

// ------------------------------------------------------------------ H2
struct RayBox {   // axis-aligned box + minimum near distance (by value)
    float xmin, ymin, zmin, xmax, ymax, zmax, min_near;
};
__device__ __forceinline__ void ray_box(float ox, float oy, float oz, float dx, float dy, float dz, const RayBox &b,
                                        float &near, float &far) {
    const float rdx = 1.0f / dx, rdy = 1.0f / dy, rdz = 1.0f / dz;
    const float ax = (b.xmin - ox) * rdx, bx = (b.xmax - ox) * rdx;
    const float ay = (b.ymin - oy) * rdy, by = (b.ymax - oy) * rdy;
    const float az = (b.zmin - oz) * rdz, bz = (b.zmax - oz) * rdz;
    near = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
    far = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    bool miss = !(far >= near);
    near = fmaxf(near, b.min_near);
    miss = miss || !(far >= near);
    if (miss) near = far = 3.4028234663852886e38f;
}
__global__ void __launch_bounds__(256) k_near_far(const float *__restrict__ rays_o, const float *__restrict__ rays_d,
                                                  int64_t N, RayBox box, float *__restrict__ nears,
                                                  float *__restrict__ fars) {
    for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < N; n += (int64_t)gridDim.x * blockDim.x) {
        float near, far;
        ray_box(rays_o[n * 3], rays_o[n * 3 + 1], rays_o[n * 3 + 2], rays_d[n * 3], rays_d[n * 3 + 1], rays_d[n * 3 + 2],
                box, near, far);
        nears[n] = near;
        fars[n] = far;
    }
}

// ------------------------------------------------------------------ H3
__global__ void __launch_bounds__(256) k_morton3d(const int32_t *__restrict__ coords, int64_t n,
                                                  uint32_t *__restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = morton3d((uint32_t)coords[i * 3], (uint32_t)coords[i * 3 + 1], (uint32_t)coords[i * 3 + 2]);
}
__global__ void __launch_bounds__(256) k_morton3d_invert(const uint32_t *__restrict__ idx, int64_t n,
                                                         int32_t *__restrict__ coords) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t v = idx[i];
        coords[i * 3] = (int32_t)compact_bits(v);
        coords[i * 3 + 1] = (int32_t)compact_bits(v >> 1);
        coords[i * 3 + 2] = (int32_t)compact_bits(v >> 2);
    }
}
// one thread per output byte; the 8 cells of a byte are two aligned float4 loads
__global__ void __launch_bounds__(256) k_packbits(const float *__restrict__ grid, int64_t n_bytes, float thresh,
                                                  const float *__restrict__ mean_dev, uint8_t *__restrict__ bits) {
    const float th = mean_dev ? fminf(thresh, *mean_dev) : thresh;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < n_bytes; b += (int64_t)gridDim.x * blockDim.x) {
        const float4 lo = reinterpret_cast<const float4 *>(grid)[b * 2];
        const float4 hi = reinterpret_cast<const float4 *>(grid)[b * 2 + 1];
        uint32_t v = 0;
        v |= (lo.x > th) ? 1u : 0u;
        v |= (lo.y > th) ? 2u : 0u;
        v |= (lo.z > th) ? 4u : 0u;
        v |= (lo.w > th) ? 8u : 0u;
        v |= (hi.x > th) ? 16u : 0u;
        v |= (hi.y > th) ? 32u : 0u;
        v |= (hi.z > th) ? 64u : 0u;
        v |= (hi.w > th) ? 128u : 0u;
        bits[b] = (uint8_t)v;
    }
}

// ------------------------------------------------------------------ H4 occupancy test
struct MarchParams {
    float bound;
    int cascade;
    int G;
    int max_steps;
    float dt_gamma;
    float dt_min;
    float dt_max;
};

// x already clamped to [-bound, bound].  Op order mirrors oracle/nerf_oracle.py::march_cell_index.
__device__ __forceinline__ bool cell_occupied(float x, float y, float z, float dt, const MarchParams &P,
                                              const uint8_t *__restrict__ bitfield) {
    int level = 0;
    float mip_bound = fminf(1.0f, P.bound);
    if (P.cascade > 1) {
        const float mx = fmaxf(fabsf(x), fmaxf(fabsf(y), fabsf(z)));
        int e0, e1;
        (void)frexpf(mx, &e0);
        (void)frexpf(dt * (0.5f * (float)P.G), &e1);
        e0 = min(P.cascade - 1, max(0, e0));
        e1 = min(P.cascade - 1, max(0, e1));
        level = max(e0, e1);
        mip_bound = fminf(ldexpf(1.0f, level), P.bound);
    }
    const float rb = 1.0f / mip_bound;
    const float halfG = 0.5f * (float)P.G;
    const float gm1 = (float)(P.G - 1);
    float ux = x * rb, uy = y * rb, uz = z * rb;
    ux = ux + 1.0f; uy = uy + 1.0f; uz = uz + 1.0f;
    ux = ux * halfG; uy = uy * halfG; uz = uz * halfG;
    const uint32_t nx = (uint32_t)(int)clampf(ux, 0.0f, gm1);
    const uint32_t ny = (uint32_t)(int)clampf(uy, 0.0f, gm1);
    const uint32_t nz = (uint32_t)(int)clampf(uz, 0.0f, gm1);
    const uint32_t idx = (uint32_t)level * (uint32_t)(P.G * P.G * P.G) + morton3d(nx, ny, nz);
    return (bitfield[idx >> 3] >> (idx & 7u)) & 1u;
}

// Per-ray jitter of the march start.  Either a table of values (the upstream `noises = torch.rand(N)`), or a
// counter-based generator: u = hash(ray, seed, *counter) in [0,1).  The counter lives on the device and is advanced
// by the scan pass of every call, so a replayed hipGraph draws fresh jitter without any host-side RNG state
// (torch.rand inside a captured graph costs three extra kernels per replay: measured 33 us per step).
struct MarchNoise {
    const float *values;     // [N] or null
    const int32_t *counter;  // device counter or null
    uint32_t seed;
    int bias;                // the write pass runs after the scan pass has advanced the counter: bias 1
};
__host__ __device__ __forceinline__ float march_hash_uniform(uint32_t ray, uint32_t seed, uint32_t step) {
    uint32_t x = ray * 0x9E3779B1u + seed;
    x ^= step * 0x85EBCA77u;
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return (float)(x >> 8) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ float march_noise(const MarchNoise &nz, int64_t n) {
    if (nz.values) return nz.values[n];
    if (!nz.counter) return 0.0f;
    return march_hash_uniform((uint32_t)n, nz.seed, (uint32_t)(*nz.counter - nz.bias));
}

// Totals of a training march from the per-ray counts, by ONE wavefront (every lane must call it): what k_march_scan
// writes to `counter`, by the same rule -- a ray is dropped when the samples of all rays before it (dropped or not) plus
// its own exceed the capacity.  64 rays per round: inclusive wave scan + running carry.  Advances the jitter counter.
#ifndef LNERF_MARCH_FUSED_MAX_RAYS
#define LNERF_MARCH_FUSED_MAX_RAYS 8192   // above: the per-wavefront prefix (N^2 / 2 loads) loses to the scan kernel
#endif
constexpr int64_t MARCH_FUSED_MAX_RAYS = LNERF_MARCH_FUSED_MAX_RAYS;
// counter[3]: running maximum, over the marches since the caller last zeroed it, of  M | (rays dropped ? 2^30 : 0)  --
// what a training loop that sizes its sample buffers from observed marches reads back once in a while
// (NeRFRenderer.update_sample_budget), kept by the ONE thread that writes the totals: no launch, no atomics
__device__ __forceinline__ void march_note_peak(int32_t *__restrict__ counter, int32_t m, int dropped) {
    const int32_t word = m | (dropped > 0 ? (1 << 30) : 0);
    const int32_t old = counter[3];
    counter[3] = word > old ? word : old;
}
__device__ __forceinline__ void march_totals(const int32_t *__restrict__ cnt, int64_t N, int64_t capacity,
                                             int32_t *__restrict__ counter, int32_t *__restrict__ noise_counter) {
    const int lane = lane_id();
    long long carry = 0, best = 0;
    int live = 0, drop = 0;
    for (int64_t base = 0; base < N; base += 64) {
        const int64_t i = base + lane;
        const int c = i < N ? cnt[i] : 0;   // (one round trip per 64 rays: this is the rare overflow path)
        long long inc = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const long long u = __shfl_up(inc, o, 64);
            if (lane >= o) inc += u;
        }
        const long long off = carry + inc - c;
        const bool dropped = (c > 0) && (off + c > capacity);
        if (c > 0 && !dropped) { ++live; best = off + c > best ? off + c : best; }
        drop += dropped ? 1 : 0;
        carry += __shfl(inc, 63, 64);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        live += __shfl_xor(live, o, 64);
        drop += __shfl_xor(drop, o, 64);
        const long long u = __shfl_xor(best, o, 64);
        best = u > best ? u : best;
    }
    if (lane == 0) {
        // M: all samples when nothing was dropped, else the end of the last kept span
        counter[0] = drop > 0 ? (int32_t)best : (int32_t)(carry > capacity ? capacity : carry);
        counter[1] = live;
        counter[2] = drop;
        march_note_peak(counter, counter[0], drop);
        if (noise_counter) *noise_counter += 1;   // (this pass took its jitter from rays[][1], not from the counter)
    }
}

// chunks of 64 lattice points whose occupancy bytes one round of the uniform-step march requests together
#ifndef LNERF_MARCH_CHUNKS
#define LNERF_MARCH_CHUNKS 4
#endif
constexpr int MARCH_CHUNKS = LNERF_MARCH_CHUNKS;
// One wavefront per ray.  Each iteration tests 64 consecutive lattice points of the ray;
// ballot + popcount gives the count (pass 1) or, with mbcnt, each sample's slot (pass 2).
template <bool WRITE, bool UNIFORM_DT>
__global__ void __launch_bounds__(256)
k_march_train(const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ nears,
              const float *__restrict__ fars, RayBox box, int clip, RayGen gen, int64_t N, const uint8_t *__restrict__ bitfield, MarchParams P,
              MarchNoise noises, float *__restrict__ xyzs, float *__restrict__ dirs,
              float *__restrict__ deltas, int32_t *__restrict__ rays, int32_t *__restrict__ cnt, int64_t capacity,
              int32_t *__restrict__ counter, int32_t *__restrict__ noise_counter) {
    const int64_t n = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;  // wave-uniform
    __shared__ int s_count[4];
    if (n >= N) {
        if (!WRITE && cnt) {   // (the workgroup's other wavefronts wait for this one at the barrier below)
            if ((threadIdx.x & 63) == 0) s_count[threadIdx.x >> 6] = 0;
            __syncthreads();
        }
        return;
    }
    const int lane = lane_id();
    // the ray: generated here by the count pass of the `_pose` form (k_get_rays' arithmetic; written out for the write
    // pass and the caller), else read
    float ro[3], rd[3];
    if (!WRITE && gen.on) {
        float fx = gen.fx, fy = gen.fy, cx = gen.cx, cy = gen.cy;
        if (gen.intr) {   // (wave-uniform: one ray per wavefront)
            const float *k = gen.intr + (n / ((int64_t)gen.H * gen.W)) * 4;
            fx = k[0]; fy = k[1]; cx = k[2]; cy = k[3];
        }
        pixel_ray(gen.c2w, gen.H, gen.W, fx, fy, cx, cy, n, ro, rd);
        if (lane == 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r) { gen.ro[n * 3 + r] = ro[r]; gen.rd[n * 3 + r] = rd[r]; }
        }
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) { ro[r] = rays_o[n * 3 + r]; rd[r] = rays_d[n * 3 + r]; }
    }
    float near, far;
    if (clip) {   // the AABB clip of lnerf_near_far_from_aabb, here: one dispatch less per view (same arithmetic)
        ray_box(ro[0], ro[1], ro[2], rd[0], rd[1], rd[2], box, near, far);
    } else {
        near = nears[n]; far = fars[n];
    }
    int count = 0;
    int64_t offset = 0;
    int budget = P.max_steps;
    float noise_in = 0.f;
    if (WRITE && cnt) {
        // Scan-free form (N <= MARCH_FUSED_MAX_RAYS): the count pass left every ray's count in `cnt`, every workgroup's
        // (4 rays') sum and number of non-empty rays in cnt[N + workgroup], and the ray's jitter in rays[n][1]; this
        // wavefront sums what lies before its ray -- what the single-workgroup scan kernel between the two passes did,
        // without that dispatch (7 us of launch latency for 2 us of work).  One batch of loads for 4096 rays.  `cnt` is
        // never modified here, so every wavefront sees the original counts whatever the others have already written.
        const int32_t *wg = cnt + N;
        const int64_t w = n >> 2;
        long long part = 0;
        int nonempty = 0;
        for (int64_t j0 = 0; j0 < w; j0 += 1024) {   // sixteen loads in flight per lane
            int v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int64_t j = j0 + k * 64 + lane;
                v[k] = j < w ? wg[j] : 0;
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) { part += v[k] & 0xFFFFFF; nonempty += v[k] >> 28; }
        }
        {   // the rays of this ray's own workgroup that come before it
            const int64_t i = 4 * w + lane;
            const int c = (lane < 4 && i < n) ? cnt[i] : 0;
            part += c;
            nonempty += c > 0 ? 1 : 0;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            part += __shfl_xor(part, o, 64);
            nonempty += __shfl_xor(nonempty, o, 64);
        }
        const int c = cnt[n];
        const bool dropped = (c > 0) && (part + c > capacity);   // (k_march_scan's rule)
        offset = dropped ? 0 : part;
        budget = dropped ? 0 : c;
        noise_in = __int_as_float(rays[n * 3 + 1]);
        if (lane == 0) {
            rays[n * 3 + 1] = (int32_t)offset;
            if (dropped) rays[n * 3 + 2] = 0;
        }
        if (n == N - 1) {   // the last ray's wavefront has the totals at hand
            const long long total = part + c;
            if (total <= capacity) {
                if (lane == 0) {
                    counter[0] = (int32_t)total;
                    counter[1] = nonempty + (c > 0 ? 1 : 0);
                    counter[2] = 0;
                    march_note_peak(counter, (int32_t)total, 0);
                    if (noise_counter) *noise_counter += 1;   // (this pass took its jitter from rays[][1])
                }
            } else {
                march_totals(cnt, N, capacity, counter, noise_counter);   // rays were dropped: the scan's bookkeeping
            }
        }
    } else if (WRITE) {
        offset = rays[n * 3 + 1];
        budget = rays[n * 3 + 2];  // 0 if the ray was dropped for capacity
    }
    if (near < far && budget > 0) {
        const float ox = ro[0], oy = ro[1], oz = ro[2];
        const float dx = rd[0], dy = rd[1], dz = rd[2];
        const float dt0 = clampf(near * P.dt_gamma, P.dt_min, P.dt_max);
        const float noise = (WRITE && cnt) ? noise_in : march_noise(noises, n);
        if (!WRITE && cnt && lane == 0) rays[n * 3 + 1] = __float_as_int(noise);   // hand the jitter to the write pass
        const float t0 = near + dt0 * noise;
        float t = t0;
        if (!UNIFORM_DT) {  // lane l starts at lattice point l
            for (int i = 0; i < lane; ++i) t = t + clampf(t * P.dt_gamma, P.dt_min, P.dt_max);
        }
        if (UNIFORM_DT) {
            // Closed-form lattice: MARCH_CHUNKS chunks of 64 lattice points per round, their occupancy bytes requested together
            // -- the pass is a chain of dependent L2 round trips (one per chunk, ~16 per ray), this makes it ~16 / MARCH_CHUNKS.  Chunks
            // past the ray's end read a clamped (valid) cell and are ignored; the decisions and their order are those
            // of the one-chunk loop below.
            const float dt = P.dt_min;
            bool done = false;
            for (int base = 0; base < (1 << 22) && !done; base += 64 * MARCH_CHUNKS) {  // bound: a non-finite `far` must not spin
                float tj[MARCH_CHUNKS], xj[MARCH_CHUNKS], yj[MARCH_CHUNKS], zj[MARCH_CHUNKS];
                bool vj[MARCH_CHUNKS], oj[MARCH_CHUNKS];
#pragma unroll
                for (int j = 0; j < MARCH_CHUNKS; ++j) {
                    float tt = (float)(base + 64 * j + lane) * dt;
                    tt = tt + t0;
                    tj[j] = tt;
                    vj[j] = tt < far;
                    float x = dx * tt, y = dy * tt, z = dz * tt;
                    x = x + ox; y = y + oy; z = z + oz;
                    xj[j] = clampf(x, -P.bound, P.bound);
                    yj[j] = clampf(y, -P.bound, P.bound);
                    zj[j] = clampf(z, -P.bound, P.bound);
                }
#pragma unroll
                for (int j = 0; j < MARCH_CHUNKS; ++j) oj[j] = cell_occupied(xj[j], yj[j], zj[j], dt, P, bitfield);
#pragma unroll
                for (int j = 0; j < MARCH_CHUNKS; ++j) {
                    if (done) break;
                    if (__ballot(vj[j]) == 0ull) { done = true; break; }  // lattice is monotone: nothing further is valid
                    const bool occ = vj[j] && oj[j];
                    const unsigned long long mask = __ballot(occ);
                    const int rank = count + mbcnt(mask);
                    if (WRITE && occ && rank < budget) {
                        const int64_t s = offset + rank;
                        // one 12-byte store per position / direction, one 8-byte store per (dt, t) pair
                        *reinterpret_cast<float3 *>(xyzs + s * 3) = make_float3(xj[j], yj[j], zj[j]);
                        *reinterpret_cast<float3 *>(dirs + s * 3) = make_float3(dx, dy, dz);
                        *reinterpret_cast<float2 *>(deltas + s * 2) = make_float2(dt, tj[j]);
                    }
                    count += __popcll(mask);
                    if (count >= budget) { count = budget; done = true; }
                }
            }
        }
        for (int base = 0; !UNIFORM_DT && base < (1 << 22); base += 64) {  // bound: a non-finite `far` must not spin
            float dt;
            if (UNIFORM_DT) {
                dt = P.dt_min;
                t = (float)(base + lane) * dt;
                t = t + t0;
            } else {
                dt = clampf(t * P.dt_gamma, P.dt_min, P.dt_max);
            }
            const bool valid = t < far;
            if (__ballot(valid) == 0ull) break;  // lattice is monotone: nothing further is valid
            float x = dx * t, y = dy * t, z = dz * t;
            x = x + ox; y = y + oy; z = z + oz;
            x = clampf(x, -P.bound, P.bound);
            y = clampf(y, -P.bound, P.bound);
            z = clampf(z, -P.bound, P.bound);
            const bool occ = valid && cell_occupied(x, y, z, dt, P, bitfield);
            const unsigned long long mask = __ballot(occ);
            const int rank = count + mbcnt(mask);
            if (WRITE && occ && rank < budget) {
                const int64_t s = offset + rank;
                xyzs[s * 3] = x; xyzs[s * 3 + 1] = y; xyzs[s * 3 + 2] = z;
                dirs[s * 3] = dx; dirs[s * 3 + 1] = dy; dirs[s * 3 + 2] = dz;
                deltas[s * 2] = dt; deltas[s * 2 + 1] = t;
            }
            count += __popcll(mask);
            if (count >= budget) { count = budget; break; }
            if (!UNIFORM_DT) {
                for (int i = 0; i < 64; ++i) t = t + clampf(t * P.dt_gamma, P.dt_min, P.dt_max);
            }
        }
    }
    if (!WRITE && lane == 0) {
        rays[n * 3] = (int32_t)n;
        if (!(cnt && near < far)) rays[n * 3 + 1] = 0;   // (scan-free form: the jitter of a marched ray sits here)
        rays[n * 3 + 2] = count;
        if (cnt) cnt[n] = count;
    }
    if (!WRITE && cnt) {   // per workgroup: samples in bits [0,24) (4 rays x <= 65536 steps), non-empty rays in [28,31)
        if (lane == 0) s_count[threadIdx.x >> 6] = count;
        __syncthreads();
        if (threadIdx.x == 0) {
            int sum = 0, ne = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) { sum += s_count[k]; ne += s_count[k] > 0 ? 1 : 0; }
            cnt[N + blockIdx.x] = sum | (ne << 28);
        }
    }
}

// Single-workgroup exclusive scan of the per-ray counts (N is a few thousand per view).
// Also counts live rays and drops rays that would overflow `capacity`.
// Every thread owns K = ceil(N / 1024) CONSECUTIVE rays: a local sum, ONE block-wide scan of the 1024 sums, then the
// thread walks its rays again with its base offset -- two barriers for any N (a chunk-of-1024 loop met at three barriers
// per chunk; the kernel is latency from end to end).
__global__ void __launch_bounds__(1024) k_march_scan(int32_t *__restrict__ rays, int64_t N, int64_t capacity,
                                                     int32_t *__restrict__ counter, int32_t *__restrict__ noise_counter) {
    __shared__ long long wave_tot[16];
    __shared__ int wave_live[16], wave_drop[16];
    __shared__ long long wave_best[16];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    // the count pass has drawn this call's jitter; advance the generator (the write pass undoes the step: bias 1)
    if (tid == 0 && noise_counter) *noise_counter += 1;
    const int64_t K = (N + 1023) / 1024;
    const int64_t n0 = (int64_t)tid * K, n1 = (n0 + K < N) ? n0 + K : N;
    long long mine = 0;
    for (int64_t n = n0; n < n1; ++n) mine += rays[n * 3 + 2];
    // inclusive scan of `mine` over the workgroup
    long long inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const long long u = __shfl_up(inc, o, 64);
        if (lane >= o) inc += u;
    }
    if (lane == 63) wave_tot[wid] = inc;
    __syncthreads();
    long long off = inc - mine;
    for (int w = 0; w < wid; ++w) off += wave_tot[w];
    int live = 0, drop = 0;
    long long best = 0;
    for (int64_t n = n0; n < n1; ++n) {
        const int c = rays[n * 3 + 2];
        const bool dropped = (c > 0) && (off + c > capacity);
        rays[n * 3 + 1] = (int32_t)(dropped ? 0 : off);
        if (dropped) rays[n * 3 + 2] = 0;
        if (c > 0 && !dropped) { ++live; best = off + c; }   // (empty rays carry a meaningless offset)
        drop += dropped ? 1 : 0;
        off += c;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        live += __shfl_xor(live, o, 64);
        drop += __shfl_xor(drop, o, 64);
        const long long u = __shfl_xor(best, o, 64);
        best = u > best ? u : best;
    }
    if (lane == 0) { wave_live[wid] = live; wave_drop[wid] = drop; wave_best[wid] = best; }
    __syncthreads();
    if (tid == 0) {
        long long total = 0, bmax = 0;
        int lv = 0, dr = 0;
        for (int w = 0; w < 16; ++w) {
            total += wave_tot[w];
            lv += wave_live[w];
            dr += wave_drop[w];
            bmax = wave_best[w] > bmax ? wave_best[w] : bmax;
        }
        // M: all samples when nothing was dropped, else the end of the last kept span
        counter[0] = dr > 0 ? (int32_t)bmax : (int32_t)(total > capacity ? capacity : total);
        counter[1] = lv;
        counter[2] = dr;
        march_note_peak(counter, counter[0], dr);
    }
}

// ------------------------------------------------------------------ H4 inference
// One thread per alive ray: emit up to n_step occupied lattice samples starting at rays_t.
__global__ void __launch_bounds__(256)
k_march_rays(int64_t n_alive, int n_step, const int32_t *__restrict__ rays_alive, const float *__restrict__ rays_t,
             const float *__restrict__ rays_o, const float *__restrict__ rays_d, const float *__restrict__ fars,
             const uint8_t *__restrict__ bitfield, MarchParams P, float *__restrict__ xyzs, float *__restrict__ dirs,
             float *__restrict__ deltas) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_alive) return;
    const int32_t n = rays_alive[i];
    float *px = xyzs + i * n_step * 3, *pd = dirs + i * n_step * 3, *pt = deltas + i * n_step * 2;
    int step = 0;
    if (n >= 0) {
        const float ox = rays_o[n * 3], oy = rays_o[n * 3 + 1], oz = rays_o[n * 3 + 2];
        const float dx = rays_d[n * 3], dy = rays_d[n * 3 + 1], dz = rays_d[n * 3 + 2];
        const float far = fars[n];
        float t = rays_t[n];
        int guard = 0;
        while (t < far && step < n_step && guard < (1 << 22)) {
            const float dt = clampf(t * P.dt_gamma, P.dt_min, P.dt_max);
            float x = dx * t, y = dy * t, z = dz * t;
            x = x + ox; y = y + oy; z = z + oz;
            x = clampf(x, -P.bound, P.bound);
            y = clampf(y, -P.bound, P.bound);
            z = clampf(z, -P.bound, P.bound);
            if (cell_occupied(x, y, z, dt, P, bitfield)) {
                px[step * 3] = x; px[step * 3 + 1] = y; px[step * 3 + 2] = z;
                pd[step * 3] = dx; pd[step * 3 + 1] = dy; pd[step * 3 + 2] = dz;
                pt[step * 2] = dt; pt[step * 2 + 1] = t;
                ++step;
            }
            t = t + dt;
            ++guard;
        }
    }
    for (; step < n_step; ++step) {  // padding: dt = 0 contributes nothing; t < 0 marks "ray exhausted"
        px[step * 3] = 0.f; px[step * 3 + 1] = 0.f; px[step * 3 + 2] = 0.f;
        pd[step * 3] = 0.f; pd[step * 3 + 1] = 0.f; pd[step * 3 + 2] = 1.f;
        pt[step * 2] = 0.f; pt[step * 2 + 1] = -1.f;
    }
}

__global__ void __launch_bounds__(256)
k_composite_rays(int64_t n_alive, int n_step, int32_t *__restrict__ rays_alive, float *__restrict__ rays_t,
                 const float *__restrict__ sigmas, const float *__restrict__ rgbs, const float *__restrict__ deltas,
                 int C, float T_thresh, float *__restrict__ weights_sum, float *__restrict__ depth,
                 float *__restrict__ image, float *__restrict__ transmittance) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_alive) return;
    const int32_t n = rays_alive[i];
    if (n < 0) return;
    const float *sg = sigmas + i * n_step, *rg = rgbs + i * n_step * C, *dl = deltas + i * n_step * 2;
    float T = transmittance[n], ws = weights_sum[n], d = depth[n];
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < C; ++c) acc[c] = image[n * C + c];
    float t_last = rays_t[n];
    bool alive = true;
    int step = 0;
    for (; step < n_step; ++step) {
        const float dt = dl[step * 2], t = dl[step * 2 + 1];
        if (t < 0.f) { alive = false; break; }  // march ran past `far`
        const float alpha = 1.0f - __expf(-sg[step] * dt);
        const float w = alpha * T;
        ws += w;
        d = fmaf(w, t, d);
        for (int c = 0; c < C; ++c) acc[c] = fmaf(w, rg[step * C + c], acc[c]);
        T *= 1.0f - alpha;
        t_last = t + dt;
        if (T < T_thresh) { alive = false; break; }
    }
    transmittance[n] = T;
    weights_sum[n] = ws;
    depth[n] = d;
    for (int c = 0; c < C; ++c) image[n * C + c] = acc[c];
    rays_t[n] = t_last;
    if (!alive) rays_alive[i] = -1;
}

// Live-ray compaction: keep entries >= 0, preserve order.  Single workgroup, wave ballot +
// mbcnt prefix, LDS carry across waves.
__global__ void __launch_bounds__(1024) k_compact_rays(const int32_t *__restrict__ in, int64_t n,
                                                       int32_t *__restrict__ out, int32_t *__restrict__ n_out) {
    __shared__ int wave_cnt[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int64_t base = 0; base < n; base += 1024) {
        const int64_t i = base + tid;
        const int32_t v = (i < n) ? in[i] : -1;
        const bool keep = v >= 0;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) wave_cnt[wid] = __popcll(m);
        __syncthreads();
        int off = carry_s;
        for (int w = 0; w < wid; ++w) off += wave_cnt[w];
        if (keep) out[off + mbcnt(m)] = v;
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wave_cnt[w];
            carry_s += tot;
        }
        __syncthreads();
    }
    if (tid == 0) *n_out = carry_s;
}

// ------------------------------------------------------------------ H10 helpers
__global__ void __launch_bounds__(256)
k_occ_cell_points(const uint32_t *__restrict__ indices, int64_t n, float mip_bound, int G,
                  const float *__restrict__ noise, float *__restrict__ xyzs) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t idx = indices ? indices[i] : (uint32_t)i;
        const float cx = (float)compact_bits(idx), cy = (float)compact_bits(idx >> 1), cz = (float)compact_bits(idx >> 2);
        const float nx = noise ? noise[i * 3] : 0.5f, ny = noise ? noise[i * 3 + 1] : 0.5f,
                    nz = noise ? noise[i * 3 + 2] : 0.5f;
        const float s = 2.0f / (float)G;
        float ux = cx + nx, uy = cy + ny, uz = cz + nz;
        ux = ux * s; uy = uy * s; uz = uz * s;
        ux = ux - 1.0f; uy = uy - 1.0f; uz = uz - 1.0f;
        xyzs[i * 3] = ux * mip_bound;
        xyzs[i * 3 + 1] = uy * mip_bound;
        xyzs[i * 3 + 2] = uz * mip_bound;
    }
}

// ---- steady-state sampling of the refresh, on the device (lnerf_occ_sample): n_rand uniformly random cells + n_rand
// cells drawn uniformly from the OCCUPIED ones, jittered points inside them.  The upstream form (torch.nonzero of the
// grid, two randint, an index, a cat, a rand) is six launches and a host synchronisation on the size of the occupied
// list; here the list is compacted in ascending cell order by two launches (deterministic: replicas of a data-parallel
// run draw the same cells), its length stays on the device, and one launch draws cells and points from a counter-based
// generator u = hash(i, seed, step, k) -- restated in oracle/nerf_oracle.py occ_sample.
constexpr int OCC_BLOCK_CELLS = 4096;   // cells per workgroup of the compaction (256 threads x 16)
__host__ __device__ __forceinline__ uint32_t occ_hash(uint32_t i, uint32_t seed, uint32_t step, uint32_t k) {
    uint32_t x = i * 0x9E3779B1u + seed;
    x ^= step * 0x85EBCA77u + k * 0xC2B2AE3Du;
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__global__ void __launch_bounds__(256) k_occ_count(const float *__restrict__ grid, int64_t n_cells, int32_t *__restrict__ counts) {
    __shared__ int wsum[4];
    const int64_t base = (int64_t)blockIdx.x * OCC_BLOCK_CELLS;
    int c = 0;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int64_t i = base + k * 256 + threadIdx.x;
        c += (i < n_cells && grid[i] > 0.f) ? 1 : 0;
    }
    c = wave_inclusive_sum_i(c);
    if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = ((wsum[0] + wsum[1]) + wsum[2]) + wsum[3];
}
__global__ void __launch_bounds__(256) k_occ_fill(const float *__restrict__ grid, int64_t n_cells,
                                                  const int32_t *__restrict__ counts, int n_blocks,
                                                  int32_t *__restrict__ list, int32_t *__restrict__ total_dev) {
    __shared__ int s_part[4], s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    // cells in earlier workgroups (and, by workgroup 0, the total)
    int part = 0;
    for (int b = tid; b < (int)blockIdx.x; b += 256) part += counts[b];
    part = wave_inclusive_sum_i(part);
    if (lane == 63) s_part[w] = part;
    __syncthreads();
    int offset = ((s_part[0] + s_part[1]) + s_part[2]) + s_part[3];
    if (blockIdx.x == 0) {
        int all = 0;
        for (int b = tid; b < n_blocks; b += 256) all += counts[b];
        all = wave_inclusive_sum_i(all);
        __syncthreads();
        if (lane == 63) s_wave[w] = all;
        __syncthreads();
        if (tid == 0) *total_dev = ((s_wave[0] + s_wave[1]) + s_wave[2]) + s_wave[3];
        __syncthreads();
    }
    const int64_t base = (int64_t)blockIdx.x * OCC_BLOCK_CELLS;
    for (int k = 0; k < 16; ++k) {   // ascending cell order: 256 consecutive cells per round
        const int64_t i = base + k * 256 + tid;
        const bool occ = i < n_cells && grid[i] > 0.f;
        const unsigned long long m = __ballot(occ);
        if (lane == 0) s_wave[w] = __popcll(m);
        __syncthreads();
        int before = 0;
        for (int ww = 0; ww < w; ++ww) before += s_wave[ww];
        const int round_total = ((s_wave[0] + s_wave[1]) + s_wave[2]) + s_wave[3];
        if (occ) list[offset + before + mbcnt(m)] = (int32_t)i;
        offset += round_total;
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256)
k_occ_draw(const int32_t *__restrict__ list, const int32_t *__restrict__ total_dev, int64_t n_cells, int64_t n_rand,
           uint32_t seed, uint32_t step, float mip_bound, int G, uint32_t *__restrict__ indices, float *__restrict__ xyzs) {
    const int total = *total_dev;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n_rand; i += (int64_t)gridDim.x * blockDim.x) {
        // STRATIFIED draws, ascending by construction: draw j of a half takes one element, uniformly, from the j-th of
        // n_rand equal strata of its population -- the cells in (Morton) order for the first half, the ascending list of
        // occupied cells for the second (a stratum of less than one element repeats its element: with fewer occupied cells
        // than draws every occupied cell is drawn total / n_rand times in a row, each time with its own jitter).  Every
        // cell keeps the marginal probability of the independent draws the upstream refresh makes (n_rand / n_cells;
        // n_rand / total), no stratum is left out by chance, and consecutive lanes evaluate NEIGHBOURING cells: the density
        // query on the candidates runs at the rate of ray samples instead of that of unrelated points (gather + MLP
        // 285 -> 195 us per refresh, the per-cell maximum 75 -> 60: measured by sorting the independent draws).
        const uint32_t h = occ_hash((uint32_t)i, seed, step, 0u);
        const unsigned long long j = (unsigned long long)(i < n_rand ? i : i - n_rand);
        const unsigned long long pop = (i >= n_rand && total > 0) ? (unsigned long long)total : (unsigned long long)n_cells;
        const unsigned long long lo = j * pop / (unsigned long long)n_rand, hi = (j + 1ull) * pop / (unsigned long long)n_rand;
        unsigned long long pick = lo + (((unsigned long long)h * (hi - lo)) >> 32);   // (hi == lo: the stratum's one element)
        pick = pick < pop ? pick : pop - 1ull;
        const uint32_t idx = (i >= n_rand && total > 0) ? (uint32_t)list[(int)pick] : (uint32_t)pick;
        indices[i] = idx;
        const float nx = (float)(occ_hash((uint32_t)i, seed, step, 1u) >> 8) * (1.0f / 16777216.0f);
        const float ny = (float)(occ_hash((uint32_t)i, seed, step, 2u) >> 8) * (1.0f / 16777216.0f);
        const float nz = (float)(occ_hash((uint32_t)i, seed, step, 3u) >> 8) * (1.0f / 16777216.0f);
        const float cx = (float)compact_bits(idx), cy = (float)compact_bits(idx >> 1), cz = (float)compact_bits(idx >> 2);
        const float s = 2.0f / (float)G;
        float ux = cx + nx, uy = cy + ny, uz = cz + nz;
        ux = ux * s; uy = uy * s; uz = uz * s;
        ux = ux - 1.0f; uy = uy - 1.0f; uz = uz - 1.0f;
        xyzs[i * 3] = ux * mip_bound;
        xyzs[i * 3 + 1] = uy * mip_bound;
        xyzs[i * 3 + 2] = uz * mip_bound;
    }
}

// Occupancy refresh, order-independent form (replicas of a data-parallel run must stay bit-identical, and the sampled
// index list may name a cell more than once): pass 1 takes the MAXIMUM of the new densities per cell into a scratch
// grid (non-negative floats order as unsigned integers: atomicMax on the bits), pass 2 applies
// grid = max(grid * decay, scratch) to the touched cells.  scratch: one uint32 per cell, all zero between calls.
__global__ void __launch_bounds__(256) k_occ_update_max(unsigned int *__restrict__ scratch,
                                                        const uint32_t *__restrict__ indices, int64_t n,
                                                        const float *__restrict__ sig) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t idx = indices ? indices[i] : (uint32_t)i;
        const float s = sig[i];
        // bit 31 marks "touched" (a density of exactly 0 must still decay the cell); densities < 0 never update
        if (s >= 0.f) atomicMax(&scratch[idx], __float_as_uint(s) | 0x80000000u);
    }
}
__global__ void __launch_bounds__(256) k_occ_update_apply(float *__restrict__ grid, unsigned int *__restrict__ scratch,
                                                          const uint32_t *__restrict__ indices, int64_t n, float decay) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t idx = indices ? indices[i] : (uint32_t)i;
        // the first entry of a cell takes its maximum (and leaves the scratch word zero for the next call); further
        // entries of the same cell find zero: exactly one thread updates the cell, whatever the order
        const unsigned int m = atomicExch(&scratch[idx], 0u);
        if (!(m & 0x80000000u)) continue;
        const float old = grid[idx];
        if (old >= 0.f) grid[idx] = fmaxf(old * decay, __uint_as_float(m & 0x7FFFFFFFu));
    }
}

// mean of max(grid, 0) over all cells, in a FIXED summation order (a float atomicAdd over the partial sums would make
// the mean -- and with it the occupancy threshold min(mean, thresh) and the bitfield -- depend on arrival order):
// OCC_MEAN_BLOCKS workgroups each sum a fixed strided slice (wave sums, then the 4 waves in order) into partial[block];
// one workgroup adds the partials in index order.
constexpr int OCC_MEAN_BLOCKS = 256;
__global__ void __launch_bounds__(256) k_occ_mean_partial(const float *__restrict__ grid, int64_t n,
                                                          float *__restrict__ partial) {
    __shared__ float s_w[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        s += fmaxf(grid[i], 0.f);
    s = wave_sum(s);
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}
// apply + mean in ONE pass over the CELLS (not the candidates): a cell whose scratch word is marked takes
// grid = max(grid * decay, maximum) and the word goes back to zero; every thread adds max(grid, 0) of its cells in the
// slice order of k_occ_mean_partial, so the mean is the one the two separate launches give, bit for bit.  Streaming
// (16 B per cell) instead of one atomicExch per candidate: 58 + 9 us -> 8 us for 2 M cells / 1 M candidates.
__global__ void __launch_bounds__(256) k_occ_apply_mean(float *__restrict__ grid, unsigned int *__restrict__ scratch,
                                                        int64_t n, float decay, float *__restrict__ partial) {
    __shared__ float s_w[4];
    float s = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float g = grid[i];
        const unsigned int m = scratch[i];
        if (m & 0x80000000u) {
            scratch[i] = 0u;
            if (g >= 0.f) {
                g = fmaxf(g * decay, __uint_as_float(m & 0x7FFFFFFFu));
                grid[i] = g;
            }
        }
        s += fmaxf(g, 0.f);
    }
    s = wave_sum(s);
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((s_w[0] + s_w[1]) + s_w[2]) + s_w[3];
}
__global__ void __launch_bounds__(64) k_occ_mean_final(const float *__restrict__ partial, int blocks, float n,
                                                       float *__restrict__ mean) {
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int b = 0; b < blocks; ++b) s += partial[b];
        *mean = s / n;
    }
}

static MarchParams make_params(float bound, int cascade, int G, int max_steps, float dt_gamma) {
    MarchParams P;
    P.bound = bound;
    P.cascade = cascade;
    P.G = G;
    P.max_steps = max_steps;
    P.dt_gamma = dt_gamma;
    P.dt_min = (float)(2.0 * 1.7320508075688772 / (double)max_steps);
    P.dt_max = (float)(2.0 * 1.7320508075688772 * (double)(1 << (cascade - 1)) / (double)G);
    return P;
}

static int grid_for(int64_t n, int block = 256, int cap = 4096) {
    int64_t g = div_up(n, block);
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace lnerf

using namespace lnerf;

extern "C" {

int lnerf_get_rays(const float *c2w, int B, int H, int W, float fx, float fy, float cx, float cy, float *rays_o,
                   float *rays_d, lnerf_stream_t stream) {
    LNERF_REQUIRE(c2w && rays_o && rays_d, "get_rays: null pointer");
    LNERF_REQUIRE(B > 0 && H > 0 && W > 0, "get_rays: B,H,W must be positive (got %d,%d,%d)", B, H, W);
    LNERF_REQUIRE(fx != 0.f && fy != 0.f, "get_rays: zero focal length");
    const int64_t total = (int64_t)B * H * W;
    hipLaunchKernelGGL(k_get_rays, dim3(grid_for(total)), dim3(256), 0, as_stream(stream), c2w, B, H, W, fx, fy, cx, cy,
                       rays_o, rays_d);
    LNERF_CHECK_LAUNCH("get_rays");
    return LNERF_OK;
}

int lnerf_near_far_from_aabb(const float *rays_o, const float *rays_d, int64_t N, float xmin, float ymin, float zmin,
                             float xmax, float ymax, float zmax, float min_near, float *nears, float *fars,
                             lnerf_stream_t stream) {
    LNERF_REQUIRE(N >= 0, "near_far_from_aabb: negative N");
    if (N == 0) return LNERF_OK;
    LNERF_REQUIRE(rays_o && rays_d && nears && fars, "near_far_from_aabb: null pointer");
    LNERF_REQUIRE(xmin <= xmax && ymin <= ymax && zmin <= zmax, "near_far_from_aabb: inverted aabb");
    const RayBox box{xmin, ymin, zmin, xmax, ymax, zmax, min_near};
    hipLaunchKernelGGL(k_near_far, dim3(grid_for(N)), dim3(256), 0, as_stream(stream), rays_o, rays_d, N, box, nears, fars);
    LNERF_CHECK_LAUNCH("near_far_from_aabb");
    return LNERF_OK;
}

int lnerf_morton3d(const int32_t *coords, int64_t n, uint32_t *indices, lnerf_stream_t stream) {
    LNERF_REQUIRE(n >= 0, "morton3d: negative n");
    if (n == 0) return LNERF_OK;
    LNERF_REQUIRE(coords && indices, "morton3d: null pointer");
    hipLaunchKernelGGL(k_morton3d, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), coords, n, indices);
    LNERF_CHECK_LAUNCH("morton3d");
    return LNERF_OK;
}

int lnerf_morton3d_invert(const uint32_t *indices, int64_t n, int32_t *coords, lnerf_stream_t stream) {
    LNERF_REQUIRE(n >= 0, "morton3d_invert: negative n");
    if (n == 0) return LNERF_OK;
    LNERF_REQUIRE(coords && indices, "morton3d_invert: null pointer");
    hipLaunchKernelGGL(k_morton3d_invert, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), indices, n, coords);
    LNERF_CHECK_LAUNCH("morton3d_invert");
    return LNERF_OK;
}

int lnerf_packbits(const float *grid, int64_t n_cells, float thresh, const float *mean_dev, uint8_t *bitfield,
                   lnerf_stream_t stream) {
    LNERF_REQUIRE(n_cells >= 0 && n_cells % 8 == 0, "packbits: n_cells (%lld) must be a multiple of 8",
                  (long long)n_cells);
    if (n_cells == 0) return LNERF_OK;
    LNERF_REQUIRE(grid && bitfield, "packbits: null pointer");
    LNERF_REQUIRE(((uintptr_t)grid & 15) == 0, "packbits: grid must be 16-byte aligned");
    hipLaunchKernelGGL(k_packbits, dim3(grid_for(n_cells / 8)), dim3(256), 0, as_stream(stream), grid, n_cells / 8,
                       thresh, mean_dev, bitfield);
    LNERF_CHECK_LAUNCH("packbits");
    return LNERF_OK;
}

static int check_march_common(const char *who, float bound, int cascade, int grid_size, int max_steps, float dt_gamma) {
    LNERF_REQUIRE(bound > 0.f, "%s: bound must be > 0", who);
    LNERF_REQUIRE(cascade >= 1 && cascade <= 8, "%s: cascade must be in [1,8] (got %d)", who, cascade);
    LNERF_REQUIRE(grid_size >= 8 && grid_size <= 1024 && (grid_size & (grid_size - 1)) == 0,
                  "%s: grid_size must be a power of two in [8,1024] (got %d)", who, grid_size);
    LNERF_REQUIRE((int64_t)cascade * grid_size * grid_size * grid_size <= (int64_t)1 << 31,
                  "%s: occupancy grid too large", who);
    LNERF_REQUIRE(max_steps >= 1 && max_steps <= 65536, "%s: max_steps out of range (%d)", who, max_steps);
    LNERF_REQUIRE(dt_gamma >= 0.f, "%s: dt_gamma must be >= 0", who);
    return LNERF_OK;
}

static int march_train_impl(const float *rays_o, const float *rays_d, const float *nears, const float *fars,
                            const RayBox *clip_box, const RayGen *raygen, int64_t N, const uint8_t *bitfield, float bound, int cascade,
                            int grid_size, int max_steps, float dt_gamma, const float *noises, uint32_t noise_seed,
                            int32_t *noise_counter, int64_t capacity, float *xyzs, float *dirs, float *deltas,
                            int32_t *rays, int32_t *counter, lnerf_stream_t stream) {
    int rc = check_march_common("march_rays_train", bound, cascade, grid_size, max_steps, dt_gamma);
    if (rc) return rc;
    LNERF_REQUIRE(N >= 0 && N <= ((int64_t)1 << 24), "march_rays_train: N out of range (%lld)", (long long)N);
    LNERF_REQUIRE(capacity >= 0 && capacity <= 0x7FFFFFFFll, "march_rays_train: capacity out of range");
    LNERF_REQUIRE(counter, "march_rays_train: null counter");
    LNERF_REQUIRE(!(noises && noise_counter), "march_rays_train: give either a noise table or a noise counter");
    if (N == 0) {
        (void)hipMemsetAsync(counter, 0, 4 * sizeof(int32_t), as_stream(stream));
        return LNERF_OK;
    }
    LNERF_REQUIRE(rays_o && rays_d && (clip_box || (nears && fars)) && bitfield && rays, "march_rays_train: null pointer");
    LNERF_REQUIRE(capacity == 0 || (xyzs && dirs && deltas), "march_rays_train: null sample buffers");
    const MarchParams P = make_params(bound, cascade, grid_size, max_steps, dt_gamma);
    const dim3 block(256), grid((unsigned)div_up(N, 4));  // 4 wavefronts (rays) per workgroup
    hipStream_t s = as_stream(stream);
    RayBox box{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int clip = clip_box ? 1 : 0;
    if (clip_box) box = *clip_box;
    RayGen gen;
    memset(&gen, 0, sizeof(gen));
    if (raygen) gen = *raygen;
    MarchNoise nz;
    nz.values = noises; nz.counter = noise_counter; nz.seed = noise_seed; nz.bias = 0;
    // up to MARCH_FUSED_MAX_RAYS rays: two dispatches (the write pass sums the counts before its ray itself; the counts
    // live behind the four totals in `counter`, see lnerf_march_counter_len); more: count, scan, write
    int32_t *cnt = N <= MARCH_FUSED_MAX_RAYS ? counter + 4 : nullptr;
    if (dt_gamma == 0.f)
        hipLaunchKernelGGL((k_march_train<false, true>), grid, block, 0, s, rays_o, rays_d, nears, fars, box, clip, gen, N,
                           bitfield, P, nz, xyzs, dirs, deltas, rays, cnt, capacity, counter, noise_counter);
    else
        hipLaunchKernelGGL((k_march_train<false, false>), grid, block, 0, s, rays_o, rays_d, nears, fars, box, clip, gen, N,
                           bitfield, P, nz, xyzs, dirs, deltas, rays, cnt, capacity, counter, noise_counter);
    LNERF_CHECK_LAUNCH("march_rays_train(count)");
    if (!cnt) {
        hipLaunchKernelGGL(k_march_scan, dim3(1), dim3(1024), 0, s, rays, N, capacity, counter, noise_counter);
        LNERF_CHECK_LAUNCH("march_rays_train(scan)");
        nz.bias = 1;
    }
    if (dt_gamma == 0.f)
        hipLaunchKernelGGL((k_march_train<true, true>), grid, block, 0, s, rays_o, rays_d, nears, fars, box, clip, gen, N,
                           bitfield, P, nz, xyzs, dirs, deltas, rays, cnt, capacity, counter, noise_counter);
    else
        hipLaunchKernelGGL((k_march_train<true, false>), grid, block, 0, s, rays_o, rays_d, nears, fars, box, clip, gen, N,
                           bitfield, P, nz, xyzs, dirs, deltas, rays, cnt, capacity, counter, noise_counter);
    LNERF_CHECK_LAUNCH("march_rays_train(write)");
    return LNERF_OK;
}

int64_t lnerf_march_counter_len(int64_t N) { return 4 + (N >= 0 && N <= MARCH_FUSED_MAX_RAYS ? N + (N + 3) / 4 : 0); }

int lnerf_march_rays_train(const float *rays_o, const float *rays_d, const float *nears, const float *fars, int64_t N,
                           const uint8_t *bitfield, float bound, int cascade, int grid_size, int max_steps,
                           float dt_gamma, const float *noises, uint32_t noise_seed, int32_t *noise_counter,
                           int64_t capacity, float *xyzs, float *dirs, float *deltas, int32_t *rays, int32_t *counter,
                           lnerf_stream_t stream) {
    return march_train_impl(rays_o, rays_d, nears, fars, nullptr, nullptr, N, bitfield, bound, cascade, grid_size, max_steps,
                            dt_gamma, noises, noise_seed, noise_counter, capacity, xyzs, dirs, deltas, rays, counter,
                            stream);
}

int lnerf_march_rays_train_aabb(const float *rays_o, const float *rays_d, float xmin, float ymin, float zmin, float xmax,
                                float ymax, float zmax, float min_near, int64_t N, const uint8_t *bitfield, float bound,
                                int cascade, int grid_size, int max_steps, float dt_gamma, const float *noises,
                                uint32_t noise_seed, int32_t *noise_counter, int64_t capacity, float *xyzs, float *dirs,
                                float *deltas, int32_t *rays, int32_t *counter, lnerf_stream_t stream) {
    LNERF_REQUIRE(xmin <= xmax && ymin <= ymax && zmin <= zmax, "march_rays_train_aabb: inverted aabb");
    const RayBox box{xmin, ymin, zmin, xmax, ymax, zmax, min_near};
    return march_train_impl(rays_o, rays_d, nullptr, nullptr, &box, nullptr, N, bitfield, bound, cascade, grid_size, max_steps,
                            dt_gamma, noises, noise_seed, noise_counter, capacity, xyzs, dirs, deltas, rays, counter,
                            stream);
}

int lnerf_march_rays_train_pose(const float *c2w, int B, int H, int W, float fx, float fy, float cx, float cy,
                                float *rays_o_out, float *rays_d_out, float xmin, float ymin, float zmin, float xmax,
                                float ymax, float zmax, float min_near, const uint8_t *bitfield, float bound, int cascade,
                                int grid_size, int max_steps, float dt_gamma, const float *noises, uint32_t noise_seed,
                                int32_t *noise_counter, int64_t capacity, float *xyzs, float *dirs, float *deltas,
                                int32_t *rays, int32_t *counter, lnerf_stream_t stream) {
    LNERF_REQUIRE(B >= 0 && H >= 1 && W >= 1 && fx != 0.f && fy != 0.f, "march_rays_train_pose: bad camera");
    LNERF_REQUIRE(xmin <= xmax && ymin <= ymax && zmin <= zmax, "march_rays_train_pose: inverted aabb");
    LNERF_REQUIRE(B == 0 || (c2w && rays_o_out && rays_d_out), "march_rays_train_pose: null pointer");
    const RayBox box{xmin, ymin, zmin, xmax, ymax, zmax, min_near};
    RayGen gen;
    gen.c2w = c2w; gen.intr = nullptr; gen.on = 1; gen.H = H; gen.W = W; gen.fx = fx; gen.fy = fy; gen.cx = cx; gen.cy = cy;
    gen.ro = rays_o_out; gen.rd = rays_d_out;
    return march_train_impl(rays_o_out, rays_d_out, nullptr, nullptr, &box, &gen, (int64_t)B * H * W, bitfield, bound,
                            cascade, grid_size, max_steps, dt_gamma, noises, noise_seed, noise_counter, capacity, xyzs,
                            dirs, deltas, rays, counter, stream);
}

int lnerf_march_rays_train_camera(const float *c2w, const float *intrinsics, int B, int H, int W, float *rays_o_out,
                                  float *rays_d_out, float xmin, float ymin, float zmin, float xmax, float ymax,
                                  float zmax, float min_near, const uint8_t *bitfield, float bound, int cascade,
                                  int grid_size, int max_steps, float dt_gamma, const float *noises, uint32_t noise_seed,
                                  int32_t *noise_counter, int64_t capacity, float *xyzs, float *dirs, float *deltas,
                                  int32_t *rays, int32_t *counter, lnerf_stream_t stream) {
    LNERF_REQUIRE(B >= 0 && H >= 1 && W >= 1, "march_rays_train_camera: bad image size");
    LNERF_REQUIRE(xmin <= xmax && ymin <= ymax && zmin <= zmax, "march_rays_train_camera: inverted aabb");
    LNERF_REQUIRE(B == 0 || (c2w && intrinsics && rays_o_out && rays_d_out), "march_rays_train_camera: null pointer");
    const RayBox box{xmin, ymin, zmin, xmax, ymax, zmax, min_near};
    RayGen gen;
    gen.c2w = c2w; gen.intr = intrinsics; gen.on = 1; gen.H = H; gen.W = W; gen.fx = gen.fy = 1.f; gen.cx = gen.cy = 0.f;
    gen.ro = rays_o_out; gen.rd = rays_d_out;
    return march_train_impl(rays_o_out, rays_d_out, nullptr, nullptr, &box, &gen, (int64_t)B * H * W, bitfield, bound,
                            cascade, grid_size, max_steps, dt_gamma, noises, noise_seed, noise_counter, capacity, xyzs,
                            dirs, deltas, rays, counter, stream);
}

int lnerf_march_rays(int64_t n_alive, int n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                     const float *rays_d, const float *fars, const uint8_t *bitfield, float bound, int cascade,
                     int grid_size, int max_steps, float dt_gamma, float *xyzs, float *dirs, float *deltas,
                     lnerf_stream_t stream) {
    int rc = check_march_common("march_rays", bound, cascade, grid_size, max_steps, dt_gamma);
    if (rc) return rc;
    LNERF_REQUIRE(n_alive >= 0 && n_step >= 1, "march_rays: bad n_alive/n_step");
    if (n_alive == 0) return LNERF_OK;
    LNERF_REQUIRE(rays_alive && rays_t && rays_o && rays_d && fars && bitfield && xyzs && dirs && deltas,
                  "march_rays: null pointer");
    const MarchParams P = make_params(bound, cascade, grid_size, max_steps, dt_gamma);
    hipLaunchKernelGGL(k_march_rays, dim3((unsigned)div_up(n_alive, 256)), dim3(256), 0, as_stream(stream), n_alive,
                       n_step, rays_alive, rays_t, rays_o, rays_d, fars, bitfield, P, xyzs, dirs, deltas);
    LNERF_CHECK_LAUNCH("march_rays");
    return LNERF_OK;
}

int lnerf_composite_rays(int64_t n_alive, int n_step, int32_t *rays_alive, float *rays_t, const float *sigmas,
                         const float *rgbs, const float *deltas, int C, float T_thresh, float *weights_sum,
                         float *depth, float *image, float *transmittance, lnerf_stream_t stream) {
    LNERF_REQUIRE(n_alive >= 0 && n_step >= 1, "composite_rays: bad n_alive/n_step");
    LNERF_REQUIRE(C >= 1 && C <= 4, "composite_rays: C must be in [1,4] (got %d)", C);
    if (n_alive == 0) return LNERF_OK;
    LNERF_REQUIRE(rays_alive && rays_t && sigmas && rgbs && deltas && weights_sum && depth && image && transmittance,
                  "composite_rays: null pointer");
    hipLaunchKernelGGL(k_composite_rays, dim3((unsigned)div_up(n_alive, 256)), dim3(256), 0, as_stream(stream), n_alive,
                       n_step, rays_alive, rays_t, sigmas, rgbs, deltas, C, T_thresh, weights_sum, depth, image,
                       transmittance);
    LNERF_CHECK_LAUNCH("composite_rays");
    return LNERF_OK;
}

int lnerf_compact_rays(const int32_t *alive_in, int64_t n, int32_t *alive_out, int32_t *n_alive_dev,
                       lnerf_stream_t stream) {
    LNERF_REQUIRE(n >= 0, "compact_rays: negative n");
    LNERF_REQUIRE(n_alive_dev, "compact_rays: null n_alive_dev");
    LNERF_REQUIRE(n == 0 || (alive_in && alive_out), "compact_rays: null pointer");
    LNERF_REQUIRE(alive_in != alive_out || n == 0, "compact_rays: in-place compaction is not supported");
    hipLaunchKernelGGL(k_compact_rays, dim3(1), dim3(1024), 0, as_stream(stream), alive_in, n, alive_out, n_alive_dev);
    LNERF_CHECK_LAUNCH("compact_rays");
    return LNERF_OK;
}

int lnerf_occ_cell_points(const uint32_t *indices, int64_t n, int cascade_level, int grid_size, float bound,
                          const float *noise, float *xyzs, lnerf_stream_t stream) {
    LNERF_REQUIRE(n >= 0 && cascade_level >= 0 && cascade_level < 8 && grid_size > 0 && bound > 0.f,
                  "occ_cell_points: bad arguments");
    if (n == 0) return LNERF_OK;
    LNERF_REQUIRE(xyzs, "occ_cell_points: null xyzs");
    const float mip_bound = fminf((float)(1 << cascade_level), bound);
    hipLaunchKernelGGL(k_occ_cell_points, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), indices, n, mip_bound,
                       grid_size, noise, xyzs);
    LNERF_CHECK_LAUNCH("occ_cell_points");
    return LNERF_OK;
}

int lnerf_occ_sample(const float *grid_level, int64_t n_cells, int cascade_level, int grid_size, float bound,
                     int64_t n_rand, uint32_t seed, uint32_t step, int32_t *scratch, uint32_t *indices, float *xyzs,
                     lnerf_stream_t stream) {
    LNERF_REQUIRE(n_cells > 0 && n_cells <= ((int64_t)1 << 31) && n_rand >= 0 && 2 * n_rand < ((int64_t)1 << 31),
                  "occ_sample: sizes out of range");
    LNERF_REQUIRE(cascade_level >= 0 && cascade_level < 8 && grid_size > 0 && bound > 0.f, "occ_sample: bad arguments");
    if (n_rand == 0) return LNERF_OK;
    LNERF_REQUIRE(grid_level && scratch && indices && xyzs, "occ_sample: null pointer");
    const int n_blocks = (int)div_up(n_cells, (int64_t)OCC_BLOCK_CELLS);
    int32_t *counts = scratch, *total = scratch + n_blocks, *list = scratch + n_blocks + 1;   // [n_blocks | 1 | n_cells]
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_occ_count, dim3((unsigned)n_blocks), dim3(256), 0, s, grid_level, n_cells, counts);
    LNERF_CHECK_LAUNCH("occ_sample(count)");
    hipLaunchKernelGGL(k_occ_fill, dim3((unsigned)n_blocks), dim3(256), 0, s, grid_level, n_cells, counts, n_blocks, list, total);
    LNERF_CHECK_LAUNCH("occ_sample(fill)");
    const float mip_bound = fminf((float)(1 << cascade_level), bound);
    hipLaunchKernelGGL(k_occ_draw, dim3(grid_for(2 * n_rand)), dim3(256), 0, s, list, total, n_cells, n_rand, seed, step,
                       mip_bound, grid_size, indices, xyzs);
    LNERF_CHECK_LAUNCH("occ_sample(draw)");
    return LNERF_OK;
}

size_t lnerf_occ_sample_scratch_bytes(int64_t n_cells) {
    if (n_cells <= 0) return 0;
    return (size_t)(div_up(n_cells, (int64_t)OCC_BLOCK_CELLS) + 1 + n_cells) * sizeof(int32_t);
}

int lnerf_occ_update(float *grid_level, const uint32_t *indices, int64_t n, const float *new_sigmas, float decay,
                     uint32_t *scratch_cells, lnerf_stream_t stream) {
    LNERF_REQUIRE(n >= 0, "occ_update: negative n");
    if (n == 0) return LNERF_OK;
    LNERF_REQUIRE(grid_level && new_sigmas && scratch_cells, "occ_update: null pointer");
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(k_occ_update_max, dim3(grid_for(n)), dim3(256), 0, s, scratch_cells, indices, n, new_sigmas);
    LNERF_CHECK_LAUNCH("occ_update(max)");
    hipLaunchKernelGGL(k_occ_update_apply, dim3(grid_for(n)), dim3(256), 0, s, grid_level, scratch_cells, indices, n, decay);
    LNERF_CHECK_LAUNCH("occ_update(apply)");
    return LNERF_OK;
}

int lnerf_occ_update_mean(float *grid_level, int64_t n_cells, const uint32_t *indices, int64_t n, const float *new_sigmas,
                          float decay, uint32_t *scratch_cells, float *mean_dev, float *scratch256,
                          lnerf_stream_t stream) {
    LNERF_REQUIRE(n >= 0 && n_cells > 0, "occ_update_mean: bad sizes");
    LNERF_REQUIRE(grid_level && scratch_cells && mean_dev && scratch256 && (n == 0 || new_sigmas),
                  "occ_update_mean: null pointer");
    hipStream_t s = as_stream(stream);
    if (n > 0) {
        hipLaunchKernelGGL(k_occ_update_max, dim3(grid_for(n)), dim3(256), 0, s, scratch_cells, indices, n, new_sigmas);
        LNERF_CHECK_LAUNCH("occ_update_mean(max)");
    }
    hipLaunchKernelGGL(k_occ_apply_mean, dim3(OCC_MEAN_BLOCKS), dim3(256), 0, s, grid_level, scratch_cells, n_cells, decay,
                       scratch256);
    LNERF_CHECK_LAUNCH("occ_update_mean(apply)");
    hipLaunchKernelGGL(k_occ_mean_final, dim3(1), dim3(64), 0, s, scratch256, OCC_MEAN_BLOCKS, (float)n_cells, mean_dev);
    LNERF_CHECK_LAUNCH("occ_update_mean(final)");
    return LNERF_OK;
}

int lnerf_occ_mean(const float *grid, int64_t n, float *mean_dev, float *scratch256, lnerf_stream_t stream) {
    LNERF_REQUIRE(n > 0 && grid && mean_dev && scratch256, "occ_mean: bad arguments");
    hipLaunchKernelGGL(k_occ_mean_partial, dim3(OCC_MEAN_BLOCKS), dim3(256), 0, as_stream(stream), grid, n, scratch256);
    LNERF_CHECK_LAUNCH("occ_mean(partial)");
    hipLaunchKernelGGL(k_occ_mean_final, dim3(1), dim3(64), 0, as_stream(stream), scratch256, OCC_MEAN_BLOCKS, (float)n,
                       mean_dev);
    LNERF_CHECK_LAUNCH("occ_mean(final)");
    return LNERF_OK;
}

}  // extern "C"
