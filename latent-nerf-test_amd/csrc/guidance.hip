// The trainer's per-step upstream gradients in ONE launch: the SEEDED SYNTHETIC
// guidance -- the stand-in for `grad = diffusion.train_step(text_z, pred)` of src/stable_diffusion.py:248-334 where no
// diffusion model is available: grad = w(t) (latents - target_dir + s * noise), t ~ U{min_step .. max_step},
// w = sqrt(a_t)(1 - a_t) (:274, :320-321) -- and the gradient of the opacity-entropy regulariser, both written in the
// renderer's own image layout [rays, C] so that `image.backward(gradient=grad)` needs no layout change.  As torch ops
// this was eleven dependent ~5 us dispatches (+ two RNG-state fills per graph replay) of a 450 us step.
// Noise and timestep come from a counter-based generator of (seed, *step_counter, element): fresh values on every replay
// of a captured graph without host RNG state (the device step counter of the optimiser advances once per step; this
// kernel only reads it).  Restated in oracle/nerf_oracle.py synthetic_guidance().
#include "common.h"

namespace lnerf {

__host__ __device__ __forceinline__ uint32_t guid_hash(uint32_t i, uint32_t seed, uint32_t step, uint32_t k) {
    uint32_t x = i * 0x9E3779B1u + seed;
    x ^= step * 0x85EBCA77u + k * 0xC2B2AE3Du;
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

// one thread per ray: image row [C], target row of the view's direction bucket, C normal deviates (Box-Muller on two
// 24-bit uniforms per pair of channels), optionally the entropy gradient of the ray's opacity
template <int C>
__global__ void __launch_bounds__(256)
k_synthetic_guidance(const float *__restrict__ image, const float *__restrict__ targets, const int32_t *__restrict__ dirs,
                     const float *__restrict__ weights, int64_t n_rays, int rays_per_view, int n_buckets, int t_lo, int n_t,
                     float noise_scale, uint32_t seed, const int32_t *__restrict__ step_dev, float *__restrict__ grad_image,
                     const float *__restrict__ ws, float ent_scale, float ent_eps, float *__restrict__ grad_ws) {
    const uint32_t step = (uint32_t)*step_dev;
    // one timestep per step (the torch form draws torch.randint(.., [1]))
    const int t = t_lo + (int)(((unsigned long long)guid_hash(0xFFFFFFFFu, seed, step, 7u) * (unsigned long long)n_t) >> 32);
    const float w = weights[t];
    const float ek = ent_scale / (float)n_rays;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rays; r += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(r / rays_per_view), px = (int)(r - (int64_t)b * rays_per_view);
        int d = dirs[b];
        d = d < 0 ? 0 : (d >= n_buckets ? n_buckets - 1 : d);
        const float *tg = targets + ((int64_t)d * rays_per_view + px) * C;
        float z[4];
#pragma unroll
        for (int c = 0; c < 4; c += 2) {
            // u1 in (0, 1], u2 in [0, 1): z = sqrt(-2 ln u1) (cos, sin)(2 pi u2)
            const float u1 = (float)((guid_hash((uint32_t)r, seed, step, (uint32_t)c) >> 8) + 1u) * (1.0f / 16777216.0f);
            const float u2 = (float)(guid_hash((uint32_t)r, seed, step, (uint32_t)c + 1u) >> 8) * (1.0f / 16777216.0f);
            const float rad = sqrtf(-2.0f * logf(u1));
            float sn, cs;
            sincosf(6.28318530717958647692f * u2, &sn, &cs);
            z[c] = rad * cs;
            z[c + 1] = rad * sn;
        }
#pragma unroll
        for (int c = 0; c < C; ++c) {
            float g = z[c] * noise_scale;
            g = g + (image[r * C + c] - tg[c]);
            grad_image[r * C + c] = g * w;
        }
        if (grad_ws) {
            const float o = ws[r];
            const bool inside = o >= ent_eps && o <= 1.0f - ent_eps;
            grad_ws[r] = inside ? ek * (log2f(1.0f - o) - log2f(o)) : 0.0f;   // (k_entropy_grad's arithmetic)
        }
    }
}

}  // namespace lnerf

using namespace lnerf;

extern "C" int lnerf_synthetic_guidance(const float *image, const float *targets, const int32_t *dirs, const float *weights,
                                        int64_t n_views, int rays_per_view, int C, int n_buckets, int t_lo, int t_hi,
                                        float noise_scale, uint32_t seed, const int32_t *step_dev, float *grad_image,
                                        const float *weights_sum, float ent_scale, float ent_eps, float *grad_weights_sum,
                                        lnerf_stream_t stream) {
    LNERF_REQUIRE(n_views >= 0 && rays_per_view > 0 && n_buckets > 0, "synthetic_guidance: bad sizes");
    LNERF_REQUIRE(C == 3 || C == 4, "synthetic_guidance: C must be 3 or 4 (got %d)", C);
    LNERF_REQUIRE(t_lo >= 0 && t_hi >= t_lo, "synthetic_guidance: need 0 <= t_lo <= t_hi");
    LNERF_REQUIRE(!grad_weights_sum || (weights_sum && ent_eps > 0.f && ent_eps < 0.5f), "synthetic_guidance: bad entropy term");
    if (n_views == 0) return LNERF_OK;
    LNERF_REQUIRE(image && targets && dirs && weights && step_dev && grad_image, "synthetic_guidance: null pointer");
    const int64_t n = n_views * rays_per_view;
    int64_t blocks = div_up(n, (int64_t)256);
    if (blocks > 2048) blocks = 2048;
#define LNERF_SG(CC)                                                                                                      \
    hipLaunchKernelGGL(k_synthetic_guidance<CC>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), image, targets, dirs, \
                       weights, n, rays_per_view, n_buckets, t_lo, t_hi - t_lo + 1, noise_scale, seed, step_dev, grad_image,   \
                       weights_sum, ent_scale, ent_eps, grad_weights_sum)
    if (C == 4) LNERF_SG(4); else LNERF_SG(3);
#undef LNERF_SG
    LNERF_CHECK_LAUNCH("synthetic_guidance");
    return LNERF_OK;
}
