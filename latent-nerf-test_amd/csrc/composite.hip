// H8/H9 front-to-back alpha compositing, training form (forward + backward).
//
// One wavefront per ray: lanes take consecutive samples of the ray's span (coalesced loads),
// transmittance comes from a log-space prefix scan of tau = sigma*dt across the 64 lanes with
// a scalar carry between 64-sample chunks, and the per-ray sums are wave reductions.
// The early stop "T < T_thresh" is wave-uniform because T is monotone along the ray.
#include "common.h"

namespace lnerf {

// one lane's sample of a 64-sample chunk (zeros past the end of the span)
template <int C>
struct Chunk {
    float sigma, dt, t, rgb[C];
};
template <int C>
__device__ __forceinline__ Chunk<C> load_chunk(const float *__restrict__ sigmas, const float *__restrict__ rgbs,
                                               const float *__restrict__ deltas, int64_t off, int cnt, int base, int lane) {
    Chunk<C> k;
    k.sigma = 0.f; k.dt = 0.f; k.t = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) k.rgb[c] = 0.f;
    const int i = base + lane;
    if (i < cnt) {
        const int64_t s = off + i;
        const float2 d = reinterpret_cast<const float2 *>(deltas)[s];
        k.dt = d.x; k.t = d.y;
        k.sigma = sigmas[s];
        if (C == 4) {
            const float4 v = reinterpret_cast<const float4 *>(rgbs)[s];
            k.rgb[0] = v.x; k.rgb[1] = v.y; k.rgb[2] = v.z; k.rgb[C - 1] = v.w;
        } else {
#pragma unroll
            for (int c = 0; c < C; ++c) k.rgb[c] = rgbs[s * C + c];
        }
    }
    return k;
}

template <int C>
__global__ void __launch_bounds__(256)
k_composite_train_fwd(const float *__restrict__ sigmas, const float *__restrict__ rgbs, const float *__restrict__ deltas,
                      const int32_t *__restrict__ rays, int64_t N, float T_thresh, const float *__restrict__ bg,
                      float *__restrict__ weights_sum, float *__restrict__ depth, float *__restrict__ image) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= N) return;
    const int lane = lane_id();
    const int64_t id = rays[r * 3];
    const int64_t off = rays[r * 3 + 1];
    const int cnt = rays[r * 3 + 2];
    float a_ws = 0.f, a_d = 0.f, a_c[C];
#pragma unroll
    for (int c = 0; c < C; ++c) a_c[c] = 0.f;
    float carry = 0.f;
    // The chunks of a ray are a chain through `carry`, but their INPUTS are not: the next chunk's sigma, (dt, t) and
    // latents are requested before this chunk is processed (the kernel is a few dependent memory round trips long,
    // nothing else), each as one load (8-byte delta pair, 16-byte latent row).
    Chunk<C> cur = load_chunk<C>(sigmas, rgbs, deltas, off, cnt, 0, lane);
    for (int base = 0; base < cnt; base += 64) {
        Chunk<C> nxt = cur;
        if (base + 64 < cnt) nxt = load_chunk<C>(sigmas, rgbs, deltas, off, cnt, base + 64, lane);
        const bool valid = base + lane < cnt;
        const float dt = cur.dt, t = cur.t;
        const float tau = cur.sigma * dt;     // (invalid lanes carry zeros)
        const float inc = wave_inclusive_sum(tau);
        const float excl = (inc - tau) + carry;
        const float T = expf(-excl);
        const float alpha = 1.0f - expf(-tau);
        const float w = (valid && T >= T_thresh) ? alpha * T : 0.f;
        a_ws += w;
        a_d = fmaf(w, t, a_d);
        if (w != 0.f) {
#pragma unroll
            for (int c = 0; c < C; ++c) a_c[c] = fmaf(w, cur.rgb[c], a_c[c]);
        }
        carry += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc), 63));
        if (expf(-carry) < T_thresh) break;  // every later sample starts below the threshold
        cur = nxt;
    }
    a_ws = wave_sum(a_ws);
    a_d = wave_sum(a_d);
#pragma unroll
    for (int c = 0; c < C; ++c) a_c[c] = wave_sum(a_c[c]);
    if (lane == 0) {
        weights_sum[id] = a_ws;
        depth[id] = a_d;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float v = a_c[c];
            if (bg) v = fmaf(1.0f - a_ws, bg[id * C + c], v);
            image[id * C + c] = v;
        }
    }
}

// d_rgbs[s][c] = di[c] * w (di == nullptr: zeros), one 16-byte store for the four latent channels
template <int C>
__device__ __forceinline__ void store_row(float *__restrict__ d_rgbs, int64_t s, const float *di, float w) {
    if (C == 4) {
        reinterpret_cast<float4 *>(d_rgbs)[s] = di ? make_float4(di[0] * w, di[1] * w, di[2] * w, di[C - 1] * w)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
        for (int c = 0; c < C; ++c) d_rgbs[s * C + c] = di ? di[c] * w : 0.f;
    }
}

template <int C>
__global__ void __launch_bounds__(256)
k_composite_train_bwd(const float *__restrict__ g_ws, const float *__restrict__ g_depth, const float *__restrict__ g_img,
                      const float *__restrict__ sigmas, const float *__restrict__ rgbs, const float *__restrict__ deltas,
                      const int32_t *__restrict__ rays, const float *__restrict__ weights_sum,
                      const float *__restrict__ depth, const float *__restrict__ image, const float *__restrict__ bg,
                      int64_t N, float T_thresh, float *__restrict__ d_sigmas, float *__restrict__ d_rgbs,
                      float *__restrict__ d_bg) {
    const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (r >= N) return;
    const int lane = lane_id();
    const int64_t id = rays[r * 3];
    const int64_t off = rays[r * 3 + 1];
    const int cnt = rays[r * 3 + 2];
    const float ws = weights_sum[id];
    const float dws = g_ws ? g_ws[id] : 0.f;
    const float ddp = g_depth ? g_depth[id] : 0.f;
    float di[C], bgc[C];
    // total = sum_k g_k w_k, from the forward outputs
    float total = fmaf(ddp, depth[id], dws * ws);
#pragma unroll
    for (int c = 0; c < C; ++c) {
        di[c] = g_img[id * C + c];
        bgc[c] = bg ? bg[id * C + c] : 0.f;
        const float fg = image[id * C + c] - (1.0f - ws) * bgc[c];  // sum_k w_k rgb_kc
        total = fmaf(di[c], fg - bgc[c] * ws, total);
    }
    if (d_bg && lane == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c) d_bg[id * C + c] = (1.0f - ws) * di[c];
    }
    float carry_tau = 0.f, carry_p = 0.f;
    bool stopped = false;  // wave-uniform
    Chunk<C> cur = load_chunk<C>(sigmas, rgbs, deltas, off, cnt, 0, lane);
    for (int base = 0; base < cnt; base += 64) {
        const int i = base + lane;
        const bool valid = i < cnt;
        const int64_t s = off + (valid ? i : 0);
        if (stopped) {  // zero-fill the tail of the span
            if (valid) {
                d_sigmas[s] = 0.f;
                store_row<C>(d_rgbs, s, nullptr, 0.f);
            }
            continue;
        }
        Chunk<C> nxt = cur;   // the next chunk's inputs, requested before this chunk's arithmetic (see the forward)
        if (base + 64 < cnt) nxt = load_chunk<C>(sigmas, rgbs, deltas, off, cnt, base + 64, lane);
        const float dt = cur.dt, t = cur.t;
        const float tau = cur.sigma * dt;
        const float inc = wave_inclusive_sum(tau);
        const float excl = (inc - tau) + carry_tau;
        const float T = expf(-excl);
        const float e = expf(-tau);
        const float alpha = 1.0f - e;
        const bool keep = valid && T >= T_thresh;
        const float w = keep ? alpha * T : 0.f;
        float g = fmaf(ddp, t, dws);
#pragma unroll
        for (int c = 0; c < C; ++c) g = fmaf(di[c], cur.rgb[c] - bgc[c], g);
        const float gw = g * w;
        const float pinc = wave_inclusive_sum(gw) + carry_p;  // inclusive prefix of g_k w_k
        if (valid) {
            // dL/dtau_i = g_i T_{i+1} - sum_{k>i} g_k w_k
            const float dtau = keep ? fmaf(g, T * e, -(total - pinc)) : 0.f;
            d_sigmas[s] = dt * dtau;
            store_row<C>(d_rgbs, s, di, w);
        }
        carry_tau += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc), 63));
        carry_p = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pinc), 63));
        if (expf(-carry_tau) < T_thresh) stopped = true;
        cur = nxt;
    }
}

}  // namespace lnerf

using namespace lnerf;

// Gradient of the opacity-entropy regulariser (the trainer's sparsity term) with respect to weights_sum, one launch:
//   L = scale * mean_i H(p_i),  p = clamp(ws, eps, 1 - eps),  H(p) = -p log2 p - (1 - p) log2(1 - p)
//   dL/dws_i = scale / N * (log2(1 - p_i) - log2 p_i)   inside the clamp, 0 outside (the clamp's subgradient)
__global__ void __launch_bounds__(256) k_entropy_grad(const float *__restrict__ ws, int64_t N, float scale, float eps,
                                                      float *__restrict__ grad) {
    const float k = scale / (float)N;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (int64_t)gridDim.x * blockDim.x) {
        const float w = ws[i];
        const bool inside = w >= eps && w <= 1.0f - eps;
        grad[i] = inside ? k * (log2f(1.0f - w) - log2f(w)) : 0.0f;
    }
}

extern "C" {

int lnerf_opacity_entropy_grad(const float *weights_sum, int64_t N, float scale, float eps, float *grad,
                               lnerf_stream_t stream) {
    LNERF_REQUIRE(N >= 0 && eps > 0.f && eps < 0.5f, "opacity_entropy_grad: bad arguments");
    if (N == 0) return LNERF_OK;
    LNERF_REQUIRE(weights_sum && grad, "opacity_entropy_grad: null pointer");
    int64_t blocks = div_up(N, (int64_t)256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_entropy_grad, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), weights_sum, N, scale, eps,
                       grad);
    LNERF_CHECK_LAUNCH("opacity_entropy_grad");
    return LNERF_OK;
}

int lnerf_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                       const int32_t *rays, int64_t N, int C, float T_thresh, const float *bg_color,
                                       float *weights_sum, float *depth, float *image, lnerf_stream_t stream) {
    LNERF_REQUIRE(N >= 0, "composite_rays_train_forward: negative N");
    LNERF_REQUIRE(C == 3 || C == 4, "composite_rays_train_forward: C must be 3 or 4 (got %d)", C);
    if (N == 0) return LNERF_OK;
    LNERF_REQUIRE(rays && weights_sum && depth && image, "composite_rays_train_forward: null pointer");
    LNERF_REQUIRE(((uintptr_t)deltas & 7) == 0 && (C != 4 || ((uintptr_t)rgbs & 15) == 0),
                  "composite_rays_train_forward: deltas must be 8-byte and (C = 4) rgbs 16-byte aligned");
    const dim3 grid((unsigned)div_up(N, 4)), block(256);
    if (C == 4)
        hipLaunchKernelGGL(k_composite_train_fwd<4>, grid, block, 0, as_stream(stream), sigmas, rgbs, deltas, rays, N,
                           T_thresh, bg_color, weights_sum, depth, image);
    else
        hipLaunchKernelGGL(k_composite_train_fwd<3>, grid, block, 0, as_stream(stream), sigmas, rgbs, deltas, rays, N,
                           T_thresh, bg_color, weights_sum, depth, image);
    LNERF_CHECK_LAUNCH("composite_rays_train_forward");
    return LNERF_OK;
}

int lnerf_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_depth,
                                        const float *grad_image, const float *sigmas, const float *rgbs,
                                        const float *deltas, const int32_t *rays, const float *weights_sum,
                                        const float *depth, const float *image, const float *bg_color, int64_t N,
                                        int C, float T_thresh, float *grad_sigmas, float *grad_rgbs, float *grad_bg,
                                        lnerf_stream_t stream) {
    LNERF_REQUIRE(N >= 0, "composite_rays_train_backward: negative N");
    LNERF_REQUIRE(C == 3 || C == 4, "composite_rays_train_backward: C must be 3 or 4 (got %d)", C);
    if (N == 0) return LNERF_OK;
    LNERF_REQUIRE(grad_image && rays && weights_sum && depth && image && grad_sigmas && grad_rgbs,
                  "composite_rays_train_backward: null pointer");
    LNERF_REQUIRE(((uintptr_t)deltas & 7) == 0 && (C != 4 || (((uintptr_t)rgbs | (uintptr_t)grad_rgbs) & 15) == 0),
                  "composite_rays_train_backward: deltas must be 8-byte and (C = 4) rgbs / grad_rgbs 16-byte aligned");
    const dim3 grid((unsigned)div_up(N, 4)), block(256);
    if (C == 4)
        hipLaunchKernelGGL(k_composite_train_bwd<4>, grid, block, 0, as_stream(stream), grad_weights_sum, grad_depth,
                           grad_image, sigmas, rgbs, deltas, rays, weights_sum, depth, image, bg_color, N, T_thresh,
                           grad_sigmas, grad_rgbs, grad_bg);
    else
        hipLaunchKernelGGL(k_composite_train_bwd<3>, grid, block, 0, as_stream(stream), grad_weights_sum, grad_depth,
                           grad_image, sigmas, rgbs, deltas, rays, weights_sum, depth, image, bg_color, N, T_thresh,
                           grad_sigmas, grad_rgbs, grad_bg);
    LNERF_CHECK_LAUNCH("composite_rays_train_backward");
    return LNERF_OK;
}

}  // extern "C"
