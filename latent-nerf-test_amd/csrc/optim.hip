// Fused Adam step (betas/eps as src/latent_paint/training/trainer.py:93-95 configures torch.optim.Adam)
// over a flat f32 parameter buffer: one pass reads p,g,m,v and writes p,m,v, optionally clears g
// and refreshes the bf16 shadow copy that the hash gather reads.  Pure HBM streaming: 16 B/lane.
#include "common.h"
#include "adam_shared.h"

namespace lnerf {

// gradient element type: f32, or bf16 (the wire format of the data-parallel all-reduce: no cast back to f32)
template <typename TG> struct Grad4;
template <> struct Grad4<float> {
    static __device__ __forceinline__ float4 load(const float *g, int64_t i) { return reinterpret_cast<const float4 *>(g)[i]; }
    static __device__ __forceinline__ void zero(float *g, int64_t i) { reinterpret_cast<float4 *>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f); }
    static __device__ __forceinline__ float load1(const float *g, int64_t t) { return g[t]; }
    static __device__ __forceinline__ void zero1(float *g, int64_t t) { g[t] = 0.f; }
};
template <> struct Grad4<uint16_t> {
    static __device__ __forceinline__ float4 load(const uint16_t *g, int64_t i) {
        const uint2 v = reinterpret_cast<const uint2 *>(g)[i];
        return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xFFFF0000u), __uint_as_float(v.y << 16),
                           __uint_as_float(v.y & 0xFFFF0000u));
    }
    static __device__ __forceinline__ void zero(uint16_t *g, int64_t i) { reinterpret_cast<uint2 *>(g)[i] = make_uint2(0u, 0u); }
    static __device__ __forceinline__ float load1(const uint16_t *g, int64_t t) { return bf16_to_f32(g[t]); }
    static __device__ __forceinline__ void zero1(uint16_t *g, int64_t t) { g[t] = 0; }
};

// parameters and moments pass through once per step: non-temporal both ways, so that they do not push the table's bf16
// shadow (what the gather reads) and the gradients out of the caches (same policy as the fused update in grid.hip)
typedef float adam_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_once(const float *base, int64_t i) {
    const adam_f4 v = __builtin_nontemporal_load(reinterpret_cast<const adam_f4 *>(base) + i);
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st_once(float *base, int64_t i, const float4 &x) {
    adam_f4 v = {x.x, x.y, x.z, x.w};
    __builtin_nontemporal_store(v, reinterpret_cast<adam_f4 *>(base) + i);
}

template <typename TG>
__global__ void __launch_bounds__(256)
k_adam(float *__restrict__ p, TG *__restrict__ g, float *__restrict__ m, float *__restrict__ v,
       uint16_t *__restrict__ shadow, int64_t n, AdamArgs a) {
    adam_bias(a);
    const int zero_grad = a.zero_grad;
    a.zero_grad = 0;  // (adam_one works on a register copy of g; the buffer is cleared below)
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        float4 P = ld_once(p, i), G = Grad4<TG>::load(g, i);
        float4 Mv = ld_once(m, i), V = ld_once(v, i);
        adam_one(P.x, G.x, Mv.x, V.x, a);
        adam_one(P.y, G.y, Mv.y, V.y, a);
        adam_one(P.z, G.z, Mv.z, V.z, a);
        adam_one(P.w, G.w, Mv.w, V.w, a);
        st_once(p, i, P);
        st_once(m, i, Mv);
        st_once(v, i, V);
        if (zero_grad) Grad4<TG>::zero(g, i);
        if (shadow) {
            uint2 s;
            s.x = (uint32_t)f32_to_bf16(P.x) | ((uint32_t)f32_to_bf16(P.y) << 16);
            s.y = (uint32_t)f32_to_bf16(P.z) | ((uint32_t)f32_to_bf16(P.w) << 16);
            reinterpret_cast<uint2 *>(shadow)[i] = s;
        }
    }
    // tail (n % 4 elements)
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && t < n) {
        float P = p[t], G = Grad4<TG>::load1(g, t), Mv = m[t], V = v[t];
        adam_one(P, G, Mv, V, a);
        p[t] = P; m[t] = Mv; v[t] = V;
        if (zero_grad) Grad4<TG>::zero1(g, t);
        if (shadow) shadow[t] = f32_to_bf16(P);
    }
}

// Many small tensors in one launch (blockIdx.y = tensor): the MLP / background parameters.
constexpr int ADAM_MULTI_MAX = 16;
struct AdamMulti {
    float *p[ADAM_MULTI_MAX], *g[ADAM_MULTI_MAX], *m[ADAM_MULTI_MAX], *v[ADAM_MULTI_MAX];
    int64_t n[ADAM_MULTI_MAX];
    float lr[ADAM_MULTI_MAX];
    const int32_t *map[ADAM_MULTI_MAX];  // optional per tensor: two positions per element in `shadow` (-1: none)
    uint16_t *shadow;                    // bf16 image the mapped tensors are mirrored into (the MLP's weight fragments)
};
// tick: the last workgroup to arrive advances the device step counter -- the separate one-thread launch of
// lnerf_adam_tick costs a whole dispatch (~4 us in a replayed graph).  step_dev[1] is the arrival counter (left at 0
// again).  Ordering: every wave reads the counter ONCE with a device-scope atomic load (the compiler can neither
// duplicate nor re-issue it later), waits until the value has RETURNED (s_waitcnt), and only then joins the workgroup
// barrier that precedes its workgroup's arrival; the arrival is a device-scope atomic, the counter is rewritten with
// device-scope atomic stores by the one workgroup whose arrival came last.  So the store to step_dev[0] is ordered
// behind every read of it in this launch without a cache write-back (a release fence would flush the XCD's L2).
__global__ void __launch_bounds__(256) k_adam_multi(AdamMulti t, AdamArgs a, int32_t *tick) {
    if (tick) {
        const int32_t step_now = __hip_atomic_load(tick, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        adam_bias_at(a, step_now);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int total = (int)(gridDim.x * gridDim.y);
            if (__hip_atomic_fetch_add(&tick[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1) {
                __hip_atomic_store(&tick[1], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&tick[0], step_now + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else {
        adam_bias(a);
    }
    const int k = blockIdx.y;
    a.lr = t.lr[k];
    float *p = t.p[k], *g = t.g[k], *m = t.m[k], *v = t.v[k];
    const int32_t *map = t.shadow ? t.map[k] : nullptr;
    const int64_t n = t.n[k];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        float P = p[i], G = g[i], Mv = m[i], V = v[i];
        adam_one(P, G, Mv, V, a);
        p[i] = P; m[i] = Mv; v[i] = V;
        if (a.zero_grad) g[i] = G;
        if (map) {   // uniform per workgroup
            const int2 at = reinterpret_cast<const int2 *>(map)[i];
            const uint16_t h = f32_to_bf16(P);
            if (at.x >= 0) t.shadow[at.x] = h;
            if (at.y >= 0) t.shadow[at.y] = h;
        }
    }
}

__global__ void k_adam_tick(int32_t *step_dev) { *step_dev += 1; }

__global__ void __launch_bounds__(256) k_cast_bf16(const float *__restrict__ src, uint16_t *__restrict__ dst, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
        const float4 P = reinterpret_cast<const float4 *>(src)[i];
        uint2 s;
        s.x = (uint32_t)f32_to_bf16(P.x) | ((uint32_t)f32_to_bf16(P.y) << 16);
        s.y = (uint32_t)f32_to_bf16(P.z) | ((uint32_t)f32_to_bf16(P.w) << 16);
        reinterpret_cast<uint2 *>(dst)[i] = s;
    }
    const int64_t t = (n4 << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (blockIdx.x == 0 && t < n) dst[t] = f32_to_bf16(src[t]);
}

}  // namespace lnerf

using namespace lnerf;

extern "C" {

int lnerf_adam_tick(int32_t *step_dev, lnerf_stream_t stream) {
    LNERF_REQUIRE(step_dev, "adam_tick: null counter");
    hipLaunchKernelGGL(k_adam_tick, dim3(1), dim3(1), 0, as_stream(stream), step_dev);
    LNERF_CHECK_LAUNCH("adam_tick");
    return LNERF_OK;
}

int lnerf_adam_step(float *p, void *g, int grad_dtype, float *m, float *v, void *shadow_bf16, int64_t n, float lr,
                    float beta1, float beta2, float eps, int step, const int32_t *step_dev, float grad_scale,
                    int zero_grad, lnerf_stream_t stream) {
    LNERF_REQUIRE(n >= 0, "adam_step: negative n");
    LNERF_REQUIRE(step_dev || step >= 1, "adam_step: step must be >= 1 (got %d)", step);
    LNERF_REQUIRE(beta1 >= 0.f && beta1 < 1.f && beta2 >= 0.f && beta2 < 1.f, "adam_step: betas must be in [0,1)");
    LNERF_REQUIRE(grad_dtype == LNERF_F32 || grad_dtype == LNERF_BF16, "adam_step: bad gradient dtype tag");
    if (n == 0) return LNERF_OK;
    LNERF_REQUIRE(p && g && m && v, "adam_step: null pointer");
    LNERF_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0,
                  "adam_step: buffers must be 16-byte aligned");
    LNERF_REQUIRE(!shadow_bf16 || ((uintptr_t)shadow_bf16 & 7) == 0, "adam_step: shadow must be 8-byte aligned");
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)(step < 1 ? 1 : step)));
    a.bc2 = (float)(1.0 - pow((double)beta2, (double)(step < 1 ? 1 : step)));
    a.grad_scale = grad_scale;
    a.zero_grad = zero_grad;
    a.step_dev = step_dev;
    int64_t blocks = div_up(div_up(n, 4), 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    if (grad_dtype == LNERF_BF16)
        hipLaunchKernelGGL(k_adam<uint16_t>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, (uint16_t *)g, m,
                           v, (uint16_t *)shadow_bf16, n, a);
    else
        hipLaunchKernelGGL(k_adam<float>, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), p, (float *)g, m, v,
                           (uint16_t *)shadow_bf16, n, a);
    LNERF_CHECK_LAUNCH("adam_step");
    return LNERF_OK;
}

int lnerf_adam_step_multi(int count, float *const *p_host, float *const *g_host, float *const *m_host,
                          float *const *v_host, const int64_t *n_host, const float *lr_host, float beta1, float beta2,
                          float eps, int step, const int32_t *step_dev, float grad_scale, int zero_grad,
                          lnerf_stream_t stream) {
    return lnerf_adam_step_multi_shadow(count, p_host, g_host, m_host, v_host, n_host, lr_host, beta1, beta2, eps, step,
                                        step_dev, grad_scale, zero_grad, nullptr, nullptr, stream);
}

int lnerf_adam_step_multi_shadow(int count, float *const *p_host, float *const *g_host, float *const *m_host,
                                 float *const *v_host, const int64_t *n_host, const float *lr_host, float beta1,
                                 float beta2, float eps, int step, const int32_t *step_dev, float grad_scale,
                                 int zero_grad, const int32_t *const *map_host, void *shadow_bf16,
                                 lnerf_stream_t stream) {
    LNERF_REQUIRE(count >= 0 && count <= ADAM_MULTI_MAX, "adam_step_multi: count must be in [0,%d]", ADAM_MULTI_MAX);
    LNERF_REQUIRE((map_host == nullptr) == (shadow_bf16 == nullptr), "adam_step_multi: maps and shadow go together");
    LNERF_REQUIRE(step_dev || step >= 1, "adam_step_multi: step must be >= 1 (got %d)", step);
    if (count == 0) return LNERF_OK;
    LNERF_REQUIRE(p_host && g_host && m_host && v_host && n_host && lr_host, "adam_step_multi: null array");
    AdamMulti t;
    int64_t nmax = 0;
    for (int k = 0; k < count; ++k) {
        LNERF_REQUIRE(p_host[k] && g_host[k] && m_host[k] && v_host[k] && n_host[k] >= 0,
                      "adam_step_multi: bad tensor %d", k);
        t.p[k] = p_host[k]; t.g[k] = g_host[k]; t.m[k] = m_host[k]; t.v[k] = v_host[k];
        t.n[k] = n_host[k]; t.lr[k] = lr_host[k];
        t.map[k] = map_host ? map_host[k] : nullptr;
        LNERF_REQUIRE(!t.map[k] || ((uintptr_t)t.map[k] & 7) == 0, "adam_step_multi: map %d must be 8-byte aligned", k);
        nmax = n_host[k] > nmax ? n_host[k] : nmax;
    }
    t.shadow = (uint16_t *)shadow_bf16;
    AdamArgs a;
    a.lr = 0.f; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)(step < 1 ? 1 : step)));
    a.bc2 = (float)(1.0 - pow((double)beta2, (double)(step < 1 ? 1 : step)));
    a.grad_scale = grad_scale;
    a.zero_grad = zero_grad & 1;
    a.step_dev = step_dev;
    const bool tick = (zero_grad & LNERF_ADAM_TICK) != 0;
    LNERF_REQUIRE(!tick || step_dev, "adam_step_multi: LNERF_ADAM_TICK needs the device step counter");
    int64_t bx = div_up(nmax, 256);
    if (bx < 1) bx = 1;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(k_adam_multi, dim3((unsigned)bx, (unsigned)count), dim3(256), 0, as_stream(stream), t, a,
                       tick ? const_cast<int32_t *>(step_dev) : nullptr);
    LNERF_CHECK_LAUNCH("adam_step_multi");
    return LNERF_OK;
}

int lnerf_cast_f32_to_bf16(const float *src, void *dst, int64_t n, lnerf_stream_t stream) {
    LNERF_REQUIRE(n >= 0, "cast_f32_to_bf16: negative n");
    if (n == 0) return LNERF_OK;
    LNERF_REQUIRE(src && dst, "cast_f32_to_bf16: null pointer");
    LNERF_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 7) == 0, "cast_f32_to_bf16: misaligned buffers");
    int64_t blocks = div_up(div_up(n, 4), 256);
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_cast_bf16, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, (uint16_t *)dst, n);
    LNERF_CHECK_LAUNCH("cast_f32_to_bf16");
    return LNERF_OK;
}

}  // extern "C"
