// Adam arithmetic shared by the optimiser kernels (optim.hip) and the scatter's fused table update (grid.hip):
// one definition, so that both paths produce bit-identical parameters.
#pragma once
#include "common.h"

namespace lnerf {

struct AdamArgs {
    float lr, beta1, beta2, eps, bc1, bc2, grad_scale;
    float inv_bc1, inv_bc2;   // 1 / bias corrections (per launch, not per element)
    int zero_grad;
    const int32_t *step_dev;  // optional device-side step counter (hipGraph replays): overrides bc1/bc2
};

// bias corrections from the device step counter (same value in every thread; a handful of SALU/VALU ops)
__device__ __forceinline__ void adam_bias_at(AdamArgs &a, int32_t step) {
    const float t = (float)step;
    a.bc1 = 1.0f - powf(a.beta1, t);
    a.bc2 = 1.0f - powf(a.beta2, t);
    a.inv_bc1 = 1.0f / a.bc1;
    a.inv_bc2 = 1.0f / a.bc2;
}
__device__ __forceinline__ void adam_bias(AdamArgs &a) {
    if (a.step_dev) {
        adam_bias_at(a, *a.step_dev);
        return;
    }
    a.inv_bc1 = 1.0f / a.bc1;
    a.inv_bc2 = 1.0f / a.bc2;
}

__device__ __forceinline__ void adam_one(float &p, float &g, float &m, float &v, const AdamArgs &a) {
    const float gs = g * a.grad_scale;
    m = fmaf(a.beta1, m, (1.0f - a.beta1) * gs);
    v = fmaf(a.beta2, v, (1.0f - a.beta2) * gs * gs);
    // bias corrections as multiplications by per-launch reciprocals, the final quotient through v_sqrt_f32 / v_rcp_f32
    // (1 ulp each): the element-wise divisions were a quarter of the fused reduce + Adam pass's vector instructions.
    // Within ~3 ulp of the update term of torch.optim.Adam, i.e. ~3e-7 * lr on the parameter.
    const float mhat = m * a.inv_bc1;
    const float vhat = v * a.inv_bc2;
    p = p - (a.lr * mhat) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vhat) + a.eps);
    if (a.zero_grad) g = 0.f;
}

// host side: bias corrections for a host-side step number (overridden on the device when step_dev is set)
static inline void adam_host_args(AdamArgs &a, float lr, float beta1, float beta2, float eps, int step,
                                  const int32_t *step_dev, float grad_scale, int zero_grad) {
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)(step < 1 ? 1 : step)));
    a.bc2 = (float)(1.0 - pow((double)beta2, (double)(step < 1 ? 1 : step)));
    a.inv_bc1 = 1.0f / a.bc1;
    a.inv_bc2 = 1.0f / a.bc2;
    a.grad_scale = grad_scale;
    a.zero_grad = zero_grad;
    a.step_dev = step_dev;
}

}  // namespace lnerf
