// H11 background net: frequency encoding (degree 6 -> 39 dims) of the ray direction, then
// 39 -> 64 (ReLU) -> C.  Per-ray work (N = 4096 per view), negligible next to the per-sample
// kernels: one thread per ray, weights staged in LDS; backward reduces the weight gradients of a
// 64-ray tile in LDS before touching global memory.
#include "common.h"

namespace lnerf {

constexpr int BG_DEG = 6, BG_IN = 3 + 3 * 2 * BG_DEG, BG_HID = 64, BG_LDE = BG_IN + 1, BG_LDH = BG_HID + 1;

__device__ __forceinline__ void bg_encode(const float *d, float *enc) {
    enc[0] = d[0]; enc[1] = d[1]; enc[2] = d[2];
#pragma unroll
    for (int k = 0; k < BG_DEG; ++k) {
        const float f = (float)(1 << k);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            enc[3 + k * 6 + c] = sinf(d[c] * f);
            enc[3 + k * 6 + 3 + c] = cosf(d[c] * f);
        }
    }
}

__global__ void __launch_bounds__(256)
k_bg_forward(const float *__restrict__ dirs, int64_t N, const float *__restrict__ w1, const float *__restrict__ b1,
             const float *__restrict__ w2, const float *__restrict__ b2, int C, float *__restrict__ out) {
    __shared__ float sW1[BG_HID * BG_IN], sB1[BG_HID], sW2[4 * BG_HID], sB2[4];
    for (int i = threadIdx.x; i < BG_HID * BG_IN; i += 256) sW1[i] = w1[i];
    for (int i = threadIdx.x; i < C * BG_HID; i += 256) sW2[i] = w2[i];
    if (threadIdx.x < BG_HID) sB1[threadIdx.x] = b1[threadIdx.x];
    if (threadIdx.x < C) sB2[threadIdx.x] = b2[threadIdx.x];
    __syncthreads();
    for (int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x; n < N; n += (int64_t)gridDim.x * 256) {
        float d[3] = {dirs[n * 3], dirs[n * 3 + 1], dirs[n * 3 + 2]};
        float enc[BG_IN];
        bg_encode(d, enc);
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) o[c] = sB2[c];
        for (int h = 0; h < BG_HID; ++h) {
            float acc = sB1[h];
#pragma unroll
            for (int i = 0; i < BG_IN; ++i) acc = fmaf(sW1[h * BG_IN + i], enc[i], acc);
            acc = fmaxf(acc, 0.f);
            for (int c = 0; c < C; ++c) o[c] = fmaf(sW2[c * BG_HID + h], acc, o[c]);
        }
        for (int c = 0; c < C; ++c) out[n * C + c] = o[c];
    }
}

__global__ void __launch_bounds__(256)
k_bg_backward(const float *__restrict__ dirs, int64_t N, const float *__restrict__ w1, const float *__restrict__ b1,
              const float *__restrict__ w2, int C, const float *__restrict__ dout, float *__restrict__ dw1,
              float *__restrict__ db1, float *__restrict__ dw2, float *__restrict__ db2) {
    __shared__ float sW1[BG_HID * BG_IN], sB1[BG_HID], sW2[4 * BG_HID];
    __shared__ float sEnc[64 * BG_LDE], sHid[64 * BG_LDH], sDh[64 * BG_LDH], sDo[64 * 4];
    for (int i = threadIdx.x; i < BG_HID * BG_IN; i += 256) sW1[i] = w1[i];
    for (int i = threadIdx.x; i < C * BG_HID; i += 256) sW2[i] = w2[i];
    if (threadIdx.x < BG_HID) sB1[threadIdx.x] = b1[threadIdx.x];
    __syncthreads();
    for (int64_t base = (int64_t)blockIdx.x * 64; base < N; base += (int64_t)gridDim.x * 64) {
        const int r = threadIdx.x;
        if (r < 64) {
            const int64_t n = base + r;
            const bool in = n < N;
            float d[3] = {in ? dirs[n * 3] : 0.f, in ? dirs[n * 3 + 1] : 0.f, in ? dirs[n * 3 + 2] : 1.f};
            float enc[BG_IN];
            bg_encode(d, enc);
            float g[4] = {0.f, 0.f, 0.f, 0.f};
            for (int c = 0; c < C; ++c) g[c] = in ? dout[n * C + c] : 0.f;
            for (int c = 0; c < 4; ++c) sDo[r * 4 + c] = g[c];
#pragma unroll
            for (int i = 0; i < BG_IN; ++i) sEnc[r * BG_LDE + i] = enc[i];
            for (int h = 0; h < BG_HID; ++h) {
                float acc = sB1[h];
#pragma unroll
                for (int i = 0; i < BG_IN; ++i) acc = fmaf(sW1[h * BG_IN + i], enc[i], acc);
                const float hid = fmaxf(acc, 0.f);
                float dh = 0.f;
                for (int c = 0; c < C; ++c) dh = fmaf(sW2[c * BG_HID + h], g[c], dh);
                sHid[r * BG_LDH + h] = hid;
                sDh[r * BG_LDH + h] = acc > 0.f ? dh : 0.f;
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < BG_HID * BG_IN; e += 256) {
            const int h = e / BG_IN, i = e - h * BG_IN;
            float s = 0.f;
            for (int rr = 0; rr < 64; ++rr) s = fmaf(sDh[rr * BG_LDH + h], sEnc[rr * BG_LDE + i], s);
            atomicAdd(&dw1[e], s);
        }
        for (int e = threadIdx.x; e < C * BG_HID; e += 256) {
            const int c = e / BG_HID, h = e - c * BG_HID;
            float s = 0.f;
            for (int rr = 0; rr < 64; ++rr) s = fmaf(sDo[rr * 4 + c], sHid[rr * BG_LDH + h], s);
            atomicAdd(&dw2[e], s);
        }
        if (threadIdx.x < BG_HID) {
            float s = 0.f;
            for (int rr = 0; rr < 64; ++rr) s += sDh[rr * BG_LDH + threadIdx.x];
            atomicAdd(&db1[threadIdx.x], s);
        } else if (threadIdx.x < BG_HID + C) {
            const int c = threadIdx.x - BG_HID;
            float s = 0.f;
            for (int rr = 0; rr < 64; ++rr) s += sDo[rr * 4 + c];
            atomicAdd(&db2[c], s);
        }
        __syncthreads();
    }
}

}  // namespace lnerf

using namespace lnerf;

extern "C" {

int lnerf_bg_forward(const float *dirs, int64_t N, const float *w1, const float *b1, const float *w2, const float *b2,
                     int C, float *out, lnerf_stream_t stream) {
    LNERF_REQUIRE(N >= 0, "bg_forward: negative N");
    LNERF_REQUIRE(C >= 1 && C <= 4, "bg_forward: C must be in [1,4] (got %d)", C);
    if (N == 0) return LNERF_OK;
    LNERF_REQUIRE(dirs && w1 && b1 && w2 && b2 && out, "bg_forward: null pointer");
    int64_t blocks = div_up(N, 256);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_bg_forward, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), dirs, N, w1, b1, w2, b2, C,
                       out);
    LNERF_CHECK_LAUNCH("bg_forward");
    return LNERF_OK;
}

int lnerf_bg_backward(const float *dirs, int64_t N, const float *w1, const float *b1, const float *w2, const float *b2,
                      int C, const float *dout, float *dw1, float *db1, float *dw2, float *db2, lnerf_stream_t stream) {
    (void)b2;
    LNERF_REQUIRE(N >= 0, "bg_backward: negative N");
    LNERF_REQUIRE(C >= 1 && C <= 4, "bg_backward: C must be in [1,4] (got %d)", C);
    if (N == 0) return LNERF_OK;
    LNERF_REQUIRE(dirs && w1 && b1 && w2 && dout && dw1 && db1 && dw2 && db2, "bg_backward: null pointer");
    int64_t blocks = div_up(N, 64);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_bg_backward, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), dirs, N, w1, b1, w2, C,
                       dout, dw1, db1, dw2, db2);
    LNERF_CHECK_LAUNCH("bg_backward");
    return LNERF_OK;
}

}  // extern "C"
