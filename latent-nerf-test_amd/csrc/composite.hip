// H8/H9 front-to-back alpha compositing, training form (forward + backward).
//
// One wavefront per ray: lanes take consecutive samples of the ray's span (coalesced loads),
// transmittance comes from a log-space prefix scan of tau = sigma*dt across the 64 lanes with
// a scalar carry between 64-sample chunks, and the per-ray sums are wave reductions.
// The early stop "T < T_thresh" is wave-uniform because T is monotone along the ray.
#include "common.h"

namespace lnerf {

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
    for (int base = 0; base < cnt; base += 64) {
        const int i = base + lane;
        const bool valid = i < cnt;
        const int64_t s = off + (valid ? i : 0);
        const float dt = valid ? deltas[s * 2] : 0.f;
        const float t = valid ? deltas[s * 2 + 1] : 0.f;
        const float tau = valid ? sigmas[s] * dt : 0.f;
        const float inc = wave_inclusive_sum(tau);
        const float excl = (inc - tau) + carry;
        const float T = expf(-excl);
        const float alpha = 1.0f - expf(-tau);
        const float w = (valid && T >= T_thresh) ? alpha * T : 0.f;
        a_ws += w;
        a_d = fmaf(w, t, a_d);
        if (w != 0.f) {
#pragma unroll
            for (int c = 0; c < C; ++c) a_c[c] = fmaf(w, rgbs[s * C + c], a_c[c]);
        }
        carry += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc), 63));
        if (expf(-carry) < T_thresh) break;  // every later sample starts below the threshold
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
    for (int base = 0; base < cnt; base += 64) {
        const int i = base + lane;
        const bool valid = i < cnt;
        const int64_t s = off + (valid ? i : 0);
        if (stopped) {  // zero-fill the tail of the span
            if (valid) {
                d_sigmas[s] = 0.f;
#pragma unroll
                for (int c = 0; c < C; ++c) d_rgbs[s * C + c] = 0.f;
            }
            continue;
        }
        const float dt = valid ? deltas[s * 2] : 0.f;
        const float t = valid ? deltas[s * 2 + 1] : 0.f;
        const float tau = valid ? sigmas[s] * dt : 0.f;
        const float inc = wave_inclusive_sum(tau);
        const float excl = (inc - tau) + carry_tau;
        const float T = expf(-excl);
        const float e = expf(-tau);
        const float alpha = 1.0f - e;
        const bool keep = valid && T >= T_thresh;
        const float w = keep ? alpha * T : 0.f;
        float g = fmaf(ddp, t, dws);
        float rgb[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            rgb[c] = valid ? rgbs[s * C + c] : 0.f;
            g = fmaf(di[c], rgb[c] - bgc[c], g);
        }
        const float gw = g * w;
        const float pinc = wave_inclusive_sum(gw) + carry_p;  // inclusive prefix of g_k w_k
        if (valid) {
            // dL/dtau_i = g_i T_{i+1} - sum_{k>i} g_k w_k
            const float dtau = keep ? fmaf(g, T * e, -(total - pinc)) : 0.f;
            d_sigmas[s] = dt * dtau;
#pragma unroll
            for (int c = 0; c < C; ++c) d_rgbs[s * C + c] = di[c] * w;
        }
        carry_tau += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc), 63));
        carry_p = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(pinc), 63));
        if (expf(-carry_tau) < T_thresh) stopped = true;
    }
}

}  // namespace lnerf

using namespace lnerf;

extern "C" {

int lnerf_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                       const int32_t *rays, int64_t N, int C, float T_thresh, const float *bg_color,
                                       float *weights_sum, float *depth, float *image, lnerf_stream_t stream) {
    LNERF_REQUIRE(N >= 0, "composite_rays_train_forward: negative N");
    LNERF_REQUIRE(C == 3 || C == 4, "composite_rays_train_forward: C must be 3 or 4 (got %d)", C);
    if (N == 0) return LNERF_OK;
    LNERF_REQUIRE(rays && weights_sum && depth && image, "composite_rays_train_forward: null pointer");
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
