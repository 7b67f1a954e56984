// H7 fused sigma/latent MLP, bf16 MFMA path (v_mfma_f32_16x16x32_bf16, f32 accumulate).
//
// "Sample on the lane" formulation: every layer computes Z^T = W · X^T, i.e. A = a 16-row tile of the
// weight matrix, B = activations with the SAMPLE on the MFMA column (lane & 15).  The C/D tile of one
// layer (4 consecutive output features in the registers of a lane, its sample on the lane) is then
// directly the B operand of the next layer: no LDS, no cross-lane traffic between layers.  The k order
// inside a k-step is permuted by that reuse (slot (q, jj) of k-step s holds feature
// phi = 32 s + 16 (jj >> 2) + 4 q + (jj & 3)); the weight fragments are built once per workgroup with
// the same permutation baked in and live in LDS in fragment order (one conflict-free ds_read_b128 per
// operand).  The same trick runs the backward data chain (dA2 -> dA1 -> dX) with transposed weights.
// Only the weight gradients, which sum over SAMPLES, need a transpose.  dZ and H tiles are staged once per
// 128-sample workgroup step as [sample][feature] bf16 images in LDS -- a lane stores the packed registers it
// already holds for the MFMA B operand, 8 bytes (4 consecutive features of its sample) at a time -- and the
// operands of dW = dZ^T (x) H^T (feature on the lane, 8 consecutive SAMPLES in the registers) are read back with
// gfx950's transposing LDS read, ds_read_b64_tr_b16 (cdna_hip_programming.md T10): no per-element conversions or
// 2-byte stores.  Each of the four waves owns a 16-row slice of every dW (no cross-wave reduction).  One f32 slab per workgroup goes to
// HBM and k_mlp_reduce_slabs (mlp.hip) sums the slabs in a fixed order: deterministic gradients.
//
// Lane maps (cdna_hip_programming.md §3): A[i = l&15][k = 8 (l>>4) + jj], B[k = 8 (l>>4) + jj][j = l&15],
// C/D[i = 4 (l>>4) + reg][j = l&15].  The index algebra is replayed on the CPU by
// tests/test_mlp_bf16_layout_emulation.py.
#include "common.h"
#include "mlp_shared.h"

namespace lnerf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// fragment slots in LDS (each slot: 64 lanes x 8 bf16 = 1 KiB)
constexpr int F_W1A = 0;    // [mt 0..3]          forward  layer 1:  W1[16mt+c][8q+jj]
constexpr int F_W2A = 4;    // [mt 0..3][s 0..1]  forward  layer 2:  W2[16mt+c][phi]
constexpr int F_W3A = 12;   // [s 0..1]           forward  layer 3:  W3[c][phi]
constexpr int F_W3T = 14;   // [mt 0..3]          backward dA2:      W3[4q+jj][16mt+c]
constexpr int F_W2T = 18;   // [mt 0..3][s 0..1]  backward dA1:      W2[phi][16mt+c]
constexpr int F_W1T = 26;   // [mt 0..1][s 0..1]  backward dX:       W1[phi][16mt+c]
constexpr int F_FWD = 14, F_ALL = 30;

constexpr int RS = 68;      // bf16 elements per row of a [128 samples][feature] staging image (64 + 4 pad: 136 B,
                            // a multiple of 8 B as the transposing read requires; 16 lanes' 8-byte stores of one
                            // column group fall on 16 different bank pairs)

__device__ __forceinline__ int phi_of(int s, int q, int jj) { return 32 * s + 16 * (jj >> 2) + 4 * q + (jj & 3); }

// build weight fragments cooperatively (all threads of the workgroup)
__device__ __forceinline__ void build_fragments(const MlpArgs &a, __bf16 *frag, int n_frag, int tid, int nthreads) {
    for (int e = tid; e < n_frag * 512; e += nthreads) {
        const int f = e >> 9, l = (e >> 3) & 63, jj = e & 7;
        const int q = l >> 4, c = l & 15;
        float v = 0.f;
        if (f < F_W2A) {
            v = a.w1[(16 * (f - F_W1A) + c) * MLP_IN + 8 * q + jj];
        } else if (f < F_W3A) {
            const int mt = (f - F_W2A) >> 1, s = (f - F_W2A) & 1;
            v = a.w2[(16 * mt + c) * MLP_HID + phi_of(s, q, jj)];
        } else if (f < F_W3T) {
            const int s = f - F_W3A;
            v = c < a.out_dim ? a.w3[c * MLP_HID + phi_of(s, q, jj)] : 0.f;
        } else if (f < F_W2T) {
            const int mt = f - F_W3T, n = 4 * q + jj;
            v = (jj < 4 && n < a.out_dim) ? a.w3[n * MLP_HID + 16 * mt + c] : 0.f;
        } else if (f < F_W1T) {
            const int mt = (f - F_W2T) >> 1, s = (f - F_W2T) & 1;
            v = a.w2[phi_of(s, q, jj) * MLP_HID + 16 * mt + c];
        } else {
            const int mt = (f - F_W1T) >> 1, s = (f - F_W1T) & 1;
            v = a.w1[phi_of(s, q, jj) * MLP_IN + 16 * mt + c];
        }
        frag[e] = (__bf16)v;
    }
}

// The fragments of a launch, built ONCE into global memory (k_mlp_build_fragments) and copied into LDS by every
// workgroup with coalesced 16-byte loads: a workgroup building its own took 28 (forward) / 60 (backward) dependent
// scattered weight loads per thread before its first tile -- a quarter of the kernels' time at ~6 tiles per workgroup.
__global__ void __launch_bounds__(256) k_mlp_build_fragments(MlpArgs a, __bf16 *out, int n_frag) {
    build_fragments(a, out, n_frag, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
}
__device__ __forceinline__ void fetch_fragments(const MlpArgs &a, __bf16 *frag, int n_frag, int tid, int nthreads) {
    if (a.frag_global) {
        const uint4 *src = reinterpret_cast<const uint4 *>(a.frag_global);
        uint4 *dst = reinterpret_cast<uint4 *>(frag);
        for (int e = tid; e < n_frag * 64; e += nthreads) dst[e] = src[e];
    } else {
        build_fragments(a, frag, n_frag, tid, nthreads);
    }
}

__device__ __forceinline__ bf16x8 ld_frag(const __bf16 *frag, int f, int lane) {
    return *reinterpret_cast<const bf16x8 *>(frag + (f * 64 + lane) * 8);
}

// B fragment of X^T for one 16-sample column tile: lane (q, c) holds features 8q..8q+7 (levels 4q..4q+3)
__device__ __forceinline__ bf16x8 load_x(const MlpArgs &a, int64_t m, bool in, int q) {
    bf16x8 x;
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = (__bf16)0.f;
    if (!in) return x;
    if (a.feat_bf16) {
        const uint32_t *f = reinterpret_cast<const uint32_t *>(a.feat);
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) w[k] = f[(int64_t)(4 * q + k) * a.level_stride + m];
        return *reinterpret_cast<bf16x8 *>(w);
    }
    const float2 *f = reinterpret_cast<const float2 *>(a.feat);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float2 v = f[(int64_t)(4 * q + k) * a.level_stride + m];
        x[2 * k] = (__bf16)v.x;
        x[2 * k + 1] = (__bf16)v.y;
    }
    return x;
}

// relu + pack two C tiles (2s, 2s+1) into the B fragment of k-step s
__device__ __forceinline__ bf16x8 pack_relu(const f32x4 &lo, const f32x4 &hi) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r[i] = (__bf16)fmaxf(lo[i], 0.f);
        r[4 + i] = (__bf16)fmaxf(hi[i], 0.f);
    }
    return r;
}
// pack d(pre-activation) = d(activation) masked by the packed activation being positive
__device__ __forceinline__ bf16x8 pack_masked(const f32x4 &lo, const f32x4 &hi, const bf16x8 &act) {
    bf16x8 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r[i] = (float)act[i] > 0.f ? (__bf16)lo[i] : (__bf16)0.f;
        r[4 + i] = (float)act[4 + i] > 0.f ? (__bf16)hi[i] : (__bf16)0.f;
    }
    return r;
}

__device__ __forceinline__ f32x4 ld_bias4(const float *b, int base) {
    return (f32x4){b[base], b[base + 1], b[base + 2], b[base + 3]};
}

// shared forward: xB[2] -> h1B[2][2], h2B[2][2] (packed, relu'd); weights from LDS fragments
__device__ __forceinline__ void forward_hidden(const __bf16 *frag, const float *sB1, const float *sB2, int lane,
                                               const bf16x8 xB[2], bf16x8 h1B[2][2], bf16x8 h2B[2][2]) {
    const int q = lane >> 4;
    f32x4 acc[4][2];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const bf16x8 w = ld_frag(frag, F_W1A + mt, lane);
        const f32x4 b = ld_bias4(sB1, 16 * mt + 4 * q);
        acc[mt][0] = MFMA32(w, xB[0], b);
        acc[mt][1] = MFMA32(w, xB[1], b);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) h1B[s][t] = pack_relu(acc[2 * s][t], acc[2 * s + 1][t]);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 b = ld_bias4(sB2, 16 * mt + 4 * q);
        acc[mt][0] = b;
        acc[mt][1] = b;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 w = ld_frag(frag, F_W2A + 2 * mt + s, lane);
            acc[mt][0] = MFMA32(w, h1B[s][0], acc[mt][0]);
            acc[mt][1] = MFMA32(w, h1B[s][1], acc[mt][1]);
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < 2; ++t) h2B[s][t] = pack_relu(acc[2 * s][t], acc[2 * s + 1][t]);
}

// ------------------------------------------------------------------ forward
// WPS: wavefronts per SIMD the register allocation aims at (4 = four workgroups per CU, 128 VGPRs)
template <int WPS>
__global__ void __launch_bounds__(256, WPS)
k_mlp_forward_bf16(MlpArgs a, float *__restrict__ sigmas, float *__restrict__ rgbs) {
    __shared__ __attribute__((aligned(16))) __bf16 frag[F_FWD * 512];
    __shared__ float sB1[MLP_HID], sB2[MLP_HID], sB3[16];
    int64_t M = a.m_host;
    if (a.m_dev) { const int64_t md = *a.m_dev; M = md < M ? md : M; }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, c = lane & 15;
    fetch_fragments(a, frag, F_FWD, tid, 256);
    if (tid < MLP_HID) { sB1[tid] = a.b1[tid]; sB2[tid] = a.b2[tid]; }
    if (tid < 16) sB3[tid] = tid < a.out_dim ? a.b3[tid] : 0.f;
    __syncthreads();
    const int nrgb = a.out_dim - 1;
    for (int64_t tile = blockIdx.x; tile * 128 < M; tile += gridDim.x) {
        const int64_t m0 = tile * 128 + w * 32;
        bf16x8 xB[2], h1B[2][2], h2B[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) xB[t] = load_x(a, m0 + 16 * t + c, m0 + 16 * t + c < M, q);
        forward_hidden(frag, sB1, sB2, lane, xB, h1B, h2B);
        const f32x4 b3 = ld_bias4(sB3, 4 * q);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 o = b3;
#pragma unroll
            for (int s = 0; s < 2; ++s) o = MFMA32(ld_frag(frag, F_W3A + s, lane), h2B[s][t], o);
            const int64_t m = m0 + 16 * t + c;  // lane holds h[4q + r] of sample m
            if (m < M) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = 4 * q + r;
                    if (n == 0) sigmas[m] = expf(o[r] + blob_of(a, m));
                    else if (n < a.out_dim) rgbs[m * nrgb + (n - 1)] = o[r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------ backward
// store the two packed halves of a B fragment (elements 0..3 <-> features fa..fa+3, elements 4..7 <-> fb..fb+3 of
// this lane's sample) into row `row` of a [sample][feature] staging image: two 8-byte stores of registers the lane
// already holds
__device__ __forceinline__ void stage_pair(__bf16 *img, int row, int fa, int fb, const bf16x8 &v) {
    const uint4 u = *reinterpret_cast<const uint4 *>(&v);
    *reinterpret_cast<uint2 *>(img + row * RS + fa) = make_uint2(u.x, u.y);
    *reinterpret_cast<uint2 *>(img + row * RS + fb) = make_uint2(u.z, u.w);
}
__device__ __forceinline__ void stage_lo(__bf16 *img, int row, int fa, const bf16x8 &v) {
    const uint4 u = *reinterpret_cast<const uint4 *>(&v);
    *reinterpret_cast<uint2 *>(img + row * RS + fa) = make_uint2(u.x, u.y);
}

// MFMA operand with the FEATURE f0 + (lane & 15) on the lane and the 8 samples 32k + 8(lane >> 4) + 0..7 in the
// registers (A[i = feature][k = sample] and B[k = sample][j = feature] alike), read from a [sample][feature] image.
// ds_read_b64_tr_b16: per group of 16 lanes, lane 4a+b supplies the address of row a, columns 4b..4b+3 of a 4 x 16
// block, and lane i receives column i of the 4 rows; two of them cover the 8 samples.  EXEC is all ones here.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 ld_tr(const __bf16 *img, int k, int f0, int lane) {
    typedef s16x4 __attribute__((address_space(3))) *lds_s16x4_p;
    const int g = lane >> 4, i = lane & 15;
    const __bf16 *p = img + (32 * k + 8 * g + (i >> 2)) * RS + f0 + 4 * (i & 3);
    union {
        struct { s16x4 lo, hi; } h;
        bf16x8 v;
    } u;
    u.h.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)p);
    u.h.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(p + 4 * RS));
    return u.v;
}

__global__ void __launch_bounds__(256, 2)
k_mlp_backward_bf16(MlpArgs a, const float *__restrict__ sigmas, const float *__restrict__ dsigmas,
                    const float *__restrict__ drgbs, float *__restrict__ dfeat, float *__restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) __bf16 frag[F_ALL * 512];
    __shared__ __attribute__((aligned(16))) __bf16 imgA[128 * RS];  // [sample][feature]: H2, then H1, then X
    __shared__ __attribute__((aligned(16))) __bf16 imgD[128 * RS];  // [sample][feature]: dZ3, then dZ2, then dZ1
    __shared__ float sB1[MLP_HID], sB2[MLP_HID];
    int64_t M = a.m_host;
    if (a.m_dev) { const int64_t md = *a.m_dev; M = md < M ? md : M; }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, c = lane & 15;
    fetch_fragments(a, frag, F_ALL, tid, 256);
    if (tid < MLP_HID) { sB1[tid] = a.b1[tid]; sB2[tid] = a.b2[tid]; }
    __syncthreads();
    const int nrgb = a.out_dim - 1;
    const float e15 = 3269017.3724721107f;  // exp(15)
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

    // this wave's slice of the weight gradients: rows 16w..16w+15 of dW2 / dW1 (+ bias column),
    // columns 16w..16w+15 of dW3; wave 0 also owns db3
    f32x4 gW2[4], gB2 = zero4, gW1[2], gB1 = zero4, gW3 = zero4, gB3 = zero4;
#pragma unroll
    for (int i = 0; i < 4; ++i) gW2[i] = zero4;
    gW1[0] = gW1[1] = zero4;

    for (int64_t tile = blockIdx.x; tile * 128 < M; tile += gridDim.x) {
        const int64_t m0 = tile * 128 + w * 32;
        bf16x8 xB[2], h1B[2][2], h2B[2][2], dzB[2][2], d3B[2];
        // ---- dZ3^T as a B fragment: slot (q, jj < 4) <-> output 4q + jj
        bool live = false;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t m = m0 + 16 * t + c;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int n = 4 * q + jj;
                float v = 0.f;
                if (jj < 4 && n < a.out_dim && m < M)
                    v = n == 0 ? dsigmas[m] * fminf(sigmas[m], e15) : drgbs[m * nrgb + (n - 1)];
                d3B[t][jj] = (__bf16)v;
                live = live || (v != 0.f);
            }
        }
        // 128 samples whose upstream gradient is exactly zero (rays past their termination point: the
        // compositing backward writes zeros there) contribute nothing to any gradient: dfeat = 0, done
        if (!__syncthreads_or(live ? 1 : 0)) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int64_t m = m0 + 16 * t + c;
                if (m < M) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        reinterpret_cast<float2 *>(dfeat)[(int64_t)(4 * q + k) * a.level_stride + m] =
                            make_float2(0.f, 0.f);
                }
            }
            continue;  // uniform for the whole workgroup
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) xB[t] = load_x(a, m0 + 16 * t + c, m0 + 16 * t + c < M, q);
        forward_hidden(frag, sB1, sB2, lane, xB, h1B, h2B);
        // ================= stage 1: dW3 += dZ3^T (x) H2^T
        __syncthreads();  // previous step's readers of imgA/imgD are done
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = 32 * w + 16 * t + c;  // this lane's sample inside the 128-sample step
#pragma unroll
            for (int s = 0; s < 2; ++s) stage_pair(imgA, row, 32 * s + 4 * q, 32 * s + 16 + 4 * q, h2B[s][t]);
            stage_lo(imgD, row, 4 * q, d3B[t]);   // outputs 4q .. 4q+3 (columns 0..15; zero beyond out_dim)
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bf16x8 dA = ld_tr(imgD, k, 0, lane);        // A[i = output][k = sample]
            const bf16x8 hB = ld_tr(imgA, k, 16 * w, lane);   // B[k = sample][j = hidden 16w + c]
            gW3 = MFMA32(dA, hB, gW3);
            if (w == 0) gB3 = MFMA32(dA, ones, gB3);
        }
        // ---- dA2 = W3^T dZ3 ; dZ2 = dA2 masked by H2 > 0
        {
            f32x4 acc[4][2];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bf16x8 wf = ld_frag(frag, F_W3T + mt, lane);
                acc[mt][0] = MFMA32(wf, d3B[0], zero4);
                acc[mt][1] = MFMA32(wf, d3B[1], zero4);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) dzB[s][t] = pack_masked(acc[2 * s][t], acc[2 * s + 1][t], h2B[s][t]);
        }
        // ================= stage 2: dW2 += dZ2^T (x) H1^T
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = 32 * w + 16 * t + c;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                stage_pair(imgA, row, 32 * s + 4 * q, 32 * s + 16 + 4 * q, h1B[s][t]);
                stage_pair(imgD, row, 32 * s + 4 * q, 32 * s + 16 + 4 * q, dzB[s][t]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bf16x8 dA = ld_tr(imgD, k, 16 * w, lane);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) gW2[nt] = MFMA32(dA, ld_tr(imgA, k, 16 * nt, lane), gW2[nt]);
            gB2 = MFMA32(dA, ones, gB2);
        }
        // ---- dA1 = W2^T dZ2 ; dZ1 = dA1 masked by H1 > 0   (dzB is overwritten by dZ1)
        {
            f32x4 acc[4][2];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                acc[mt][0] = acc[mt][1] = zero4;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 wf = ld_frag(frag, F_W2T + 2 * mt + s, lane);
                    acc[mt][0] = MFMA32(wf, dzB[s][0], acc[mt][0]);
                    acc[mt][1] = MFMA32(wf, dzB[s][1], acc[mt][1]);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < 2; ++t) dzB[s][t] = pack_masked(acc[2 * s][t], acc[2 * s + 1][t], h1B[s][t]);
        }
        // ================= stage 3: dW1 += dZ1^T (x) X^T
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = 32 * w + 16 * t + c;
            stage_pair(imgA, row, 8 * q, 8 * q + 4, xB[t]);   // input features 8q .. 8q+7
#pragma unroll
            for (int s = 0; s < 2; ++s) stage_pair(imgD, row, 32 * s + 4 * q, 32 * s + 16 + 4 * q, dzB[s][t]);
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const bf16x8 dA = ld_tr(imgD, k, 16 * w, lane);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) gW1[nt] = MFMA32(dA, ld_tr(imgA, k, 16 * nt, lane), gW1[nt]);
            gB1 = MFMA32(dA, ones, gB1);
        }
        // ---- dX = W1^T dZ1 -> dfeat (level-major f32): lane holds features 16mt + 4q + r of its sample
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 ax[2] = {zero4, zero4};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 wf = ld_frag(frag, F_W1T + 2 * mt + s, lane);
                ax[0] = MFMA32(wf, dzB[s][0], ax[0]);
                ax[1] = MFMA32(wf, dzB[s][1], ax[1]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int64_t m = m0 + 16 * t + c;
                if (m < M) {
                    const int lv = 8 * mt + 2 * q;  // features 16mt+4q+{0,1} = level lv, {2,3} = level lv+1
                    reinterpret_cast<float2 *>(dfeat)[(int64_t)lv * a.level_stride + m] = make_float2(ax[t][0], ax[t][1]);
                    reinterpret_cast<float2 *>(dfeat)[(int64_t)(lv + 1) * a.level_stride + m] =
                        make_float2(ax[t][2], ax[t][3]);
                }
            }
        }
    }

    // ---- one slab per workgroup, every wave writes the rows it owns (layout: mlp_shared.h)
    float *slab = slabs + (int64_t)blockIdx.x * MLP_SLAB;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int o = 16 * w + 4 * q + r;  // row of dW2 / dW1
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) slab[MLP_SL_W2 + o * MLP_HID + 16 * nt + c] = gW2[nt][r];
        slab[MLP_SL_W1 + o * MLP_IN + c] = gW1[0][r];
        slab[MLP_SL_W1 + o * MLP_IN + 16 + c] = gW1[1][r];
        if (c == 0) { slab[MLP_SL_B2 + o] = gB2[r]; slab[MLP_SL_B1 + o] = gB1[r]; }
        // dW3: rows n3 = 4q + r, columns (hidden) 16w + c
        slab[MLP_SL_W3 + (4 * q + r) * MLP_HID + 16 * w + c] = gW3[r];
        if (w == 0 && c == 0) slab[MLP_SL_B3 + 4 * q + r] = gB3[r];
    }
}

}  // namespace lnerf

namespace lnerf {

int launch_mlp_fragments_bf16(const MlpArgs &a, void *frag_out, bool backward_too, hipStream_t stream) {
    const int n = backward_too ? F_ALL : F_FWD;
    static_assert((size_t)F_ALL * 1024 <= MLP_FRAG_BYTES, "fragment cache");
    hipLaunchKernelGGL(k_mlp_build_fragments, dim3((unsigned)(n * 2)), dim3(256), 0, stream, a, (__bf16 *)frag_out, n);
    LNERF_CHECK_LAUNCH("mlp(fragments)");
    return LNERF_OK;
}

int launch_mlp_forward_bf16(const MlpArgs &a, float *sigmas, float *rgbs, int blocks, int wps, hipStream_t stream) {
    if (wps >= 4) hipLaunchKernelGGL(k_mlp_forward_bf16<4>, dim3((unsigned)blocks), dim3(256), 0, stream, a, sigmas, rgbs);
    else hipLaunchKernelGGL(k_mlp_forward_bf16<2>, dim3((unsigned)blocks), dim3(256), 0, stream, a, sigmas, rgbs);
    LNERF_CHECK_LAUNCH("mlp_forward(bf16)");
    return LNERF_OK;
}

int launch_mlp_backward_bf16(const MlpArgs &a, const float *sigmas, const float *dsigmas, const float *drgbs,
                             float *dfeat, float *slabs, int blocks, hipStream_t stream) {
    hipLaunchKernelGGL(k_mlp_backward_bf16, dim3((unsigned)blocks), dim3(256), 0, stream, a, sigmas, dsigmas, drgbs,
                       dfeat, slabs);
    LNERF_CHECK_LAUNCH("mlp_backward(bf16)");
    return LNERF_OK;
}

}  // namespace lnerf
