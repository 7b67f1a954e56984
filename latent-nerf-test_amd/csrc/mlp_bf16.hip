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

// Where element e of the weight-fragment image comes from: tensor 1 / 2 / 3 (= w1 / w2 / w3) and the index into it, or
// 0 for a constant zero (padding of the 16-row output tile).  ONE statement of the layout: the builder below reads
// through it, k_mlp_fragment_maps inverts it for the optimiser's fragment shadow.
__device__ __forceinline__ int frag_source(int e, int out_dim, int &idx) {
    const int f = e >> 9, l = (e >> 3) & 63, jj = e & 7;
    const int q = l >> 4, c = l & 15;
    if (f < F_W2A) {
        idx = (16 * (f - F_W1A) + c) * MLP_IN + 8 * q + jj;
        return 1;
    } else if (f < F_W3A) {
        const int mt = (f - F_W2A) >> 1, s = (f - F_W2A) & 1;
        idx = (16 * mt + c) * MLP_HID + phi_of(s, q, jj);
        return 2;
    } else if (f < F_W3T) {
        const int s = f - F_W3A;
        idx = c * MLP_HID + phi_of(s, q, jj);
        return c < out_dim ? 3 : 0;
    } else if (f < F_W2T) {
        const int mt = f - F_W3T, n = 4 * q + jj;
        idx = n * MLP_HID + 16 * mt + c;
        return (jj < 4 && n < out_dim) ? 3 : 0;
    } else if (f < F_W1T) {
        const int mt = (f - F_W2T) >> 1, s = (f - F_W2T) & 1;
        idx = phi_of(s, q, jj) * MLP_HID + 16 * mt + c;
        return 2;
    }
    const int mt = (f - F_W1T) >> 1, s = (f - F_W1T) & 1;
    idx = phi_of(s, q, jj) * MLP_IN + 16 * mt + c;
    return 1;
}

// build weight fragments cooperatively (all threads of the workgroup)
__device__ __forceinline__ void build_fragments(const MlpArgs &a, __bf16 *frag, int n_frag, int tid, int nthreads) {
    for (int e = tid; e < n_frag * 512; e += nthreads) {
        int idx;
        const int t = frag_source(e, a.out_dim, idx);
        const float v = t == 1 ? a.w1[idx] : t == 2 ? a.w2[idx] : t == 3 ? a.w3[idx] : 0.f;
        frag[e] = (__bf16)v;
    }
}

// Inverse of the layout: map_t[2 i] / map_t[2 i + 1] = the element of the fragment image that holds weight i of tensor
// t in the forward / the transposed (backward) fragments -- every weight sits in exactly one of each.  With these the
// optimiser's multi-tensor Adam launch writes the bf16 fragments itself (lnerf_adam_step_multi_shadow), and the
// per-step fragment build (one dispatch) goes away.
__global__ void __launch_bounds__(256) k_mlp_fragment_maps(int out_dim, int32_t *m1, int32_t *m2, int32_t *m3) {
    for (int e = blockIdx.x * 256 + threadIdx.x; e < F_ALL * 512; e += gridDim.x * 256) {
        int idx;
        const int t = frag_source(e, out_dim, idx);
        const int slot = (e >> 9) < F_W3T ? 0 : 1;
        int32_t *m = t == 1 ? m1 : t == 2 ? m2 : t == 3 ? m3 : nullptr;
        if (m) m[2 * idx + slot] = e;
    }
}

// The fragments of a launch, built ONCE into global memory (k_mlp_build_fragments) and copied into LDS by every
// workgroup with coalesced 16-byte loads: a workgroup building its own took 28 (forward) / 60 (backward) dependent
// scattered weight loads per thread before its first tile -- a quarter of the kernels' time at ~6 tiles per workgroup.
__device__ __forceinline__ void build_selectors(__bf16 *frag, int tid, int nthreads);
__global__ void __launch_bounds__(256) k_mlp_build_fragments(MlpArgs a, __bf16 *out, int n_frag) {
    build_fragments(a, out, n_frag < F_ALL ? n_frag : F_ALL, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
    if (n_frag > F_ALL) build_selectors(out, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256);
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
// `m` is CLAMPED into the valid samples by the callers (no exec-masked branch around the loads): a row past the end
// reads the last sample's features, which nothing consumes -- its outputs are not stored (forward) and its upstream
// gradient is zero (backward).
__device__ __forceinline__ bf16x8 load_x(const MlpArgs &a, int64_t m, int q) {
    if (a.feat_bf16) {
        // 32-bit byte offsets from the (uniform) base: `global_load_dword v, v_off, s[base]` instead of three 64-bit
        // VALU operations per address (the launcher checks 16 * level_stride * 8 < 2^32)
        const char *f = reinterpret_cast<const char *>(a.feat);
        const uint32_t ls = (uint32_t)a.level_stride;
        uint32_t w[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            w[k] = *reinterpret_cast<const uint32_t *>(f + (size_t)((((uint32_t)(4 * q + k)) * ls + (uint32_t)m) << 2));
        return *reinterpret_cast<bf16x8 *>(w);
    }
    bf16x8 x;
    const float2 *f = reinterpret_cast<const float2 *>(a.feat);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float2 v = f[(int64_t)(4 * q + k) * a.level_stride + m];
        x[2 * k] = (__bf16)v.x;
        x[2 * k + 1] = (__bf16)v.y;
    }
    return x;
}
__device__ __forceinline__ int64_t clamp_row(int64_t m, int64_t M) { return m < M ? m : (M > 0 ? M - 1 : 0); }

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// v_cvt_pk_bf16_f32, left to the compiler: an inline-asm form hides the MFMA-result read from its hazard recogniser
__device__ __forceinline__ uint32_t cvt_pk(float lo, float hi) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f, bf16x2));
}
// relu on a packed bf16 pair: the sign bit of a bf16 is the sign bit of the int16 holding it
__device__ __forceinline__ uint32_t relu_pk(uint32_t v) {
    const s16x2 z = {0, 0};
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s16x2, v), z));
}
// 0xFFFF in every half of `act` (a relu'd pair: +0 or positive) that is non-zero: 0 - a has its sign bit set exactly
// for a in 1 .. 0x7FFF, an arithmetic shift spreads it (v_pk_sub_i16 + v_pk_ashrrev_i16; a min/sub form was turned
// into two compares, two selects and a permute per pair by the compiler)
__device__ __forceinline__ uint32_t live_pk(uint32_t act) {
    asm("" : "+v"(act));   // (opaque to the optimiser: knowing act >= 0 it rewrites the two packed ops into selects again)
    const s16x2 z = {0, 0};
    const s16x2 n = z - __builtin_bit_cast(s16x2, act);
    return __builtin_bit_cast(uint32_t, n >> 15);
}
union Pk8 {
    bf16x8 v;
    uint32_t u[4];
};
__device__ __forceinline__ bf16x8 pack_relu_pk(const f32x4 &lo, const f32x4 &hi) {
    Pk8 r;
    r.u[0] = relu_pk(cvt_pk(lo[0], lo[1]));
    r.u[1] = relu_pk(cvt_pk(lo[2], lo[3]));
    r.u[2] = relu_pk(cvt_pk(hi[0], hi[1]));
    r.u[3] = relu_pk(cvt_pk(hi[2], hi[3]));
    return r.v;
}
__device__ __forceinline__ bf16x8 pack_masked_pk(const f32x4 &lo, const f32x4 &hi, const bf16x8 &act) {
    Pk8 r, a;
    a.v = act;
    r.u[0] = cvt_pk(lo[0], lo[1]) & live_pk(a.u[0]);
    r.u[1] = cvt_pk(lo[2], lo[3]) & live_pk(a.u[1]);
    r.u[2] = cvt_pk(hi[0], hi[1]) & live_pk(a.u[2]);
    r.u[3] = cvt_pk(hi[2], hi[3]) & live_pk(a.u[3]);
    return r.v;
}
__device__ __forceinline__ bf16x8 pack_plain_pk(const f32x4 &lo, const f32x4 &hi) {
    Pk8 r;
    r.u[0] = cvt_pk(lo[0], lo[1]);
    r.u[1] = cvt_pk(lo[2], lo[3]);
    r.u[2] = cvt_pk(hi[0], hi[1]);
    r.u[3] = cvt_pk(hi[2], hi[3]);
    return r.v;
}
// sum of the 8 bf16 of a fragment, added to acc (v_dot2c_f32_bf16 against ones)
__device__ __forceinline__ float sum8(const bf16x8 &v, float acc) {
    Pk8 a;
    a.v = v;
    bf16x2 ones;
    ones[0] = (__bf16)1.0f;
    ones[1] = (__bf16)1.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a.u[i]), ones, acc, false);
    return acc;
}

// relu + pack two C tiles (2s, 2s+1) into the B fragment of k-step s
__device__ __forceinline__ bf16x8 pack_relu(const f32x4 &lo, const f32x4 &hi) { return pack_relu_pk(lo, hi); }
// pack d(pre-activation) = d(activation) masked by the packed activation being positive
__device__ __forceinline__ bf16x8 pack_masked(const f32x4 &lo, const f32x4 &hi, const bf16x8 &act) {
    return pack_masked_pk(lo, hi, act);
}

__device__ __forceinline__ f32x4 ld_bias4(const float *b, int base) {
    return (f32x4){b[base], b[base + 1], b[base + 2], b[base + 3]};
}

// Phase stamps (diagnostic builds only: -DLNERF_STAMPS, tools/mlp_stamps.py): the shader clock at the marks of ONE
// wavefront per workgroup, summed per phase into a global table.  No memory drain at a mark: what a phase waits for
// shows up in the phase that consumes it.
#ifdef LNERF_STAMPS
__device__ unsigned long long g_mlp_stamps[32];
struct Stamps {
    unsigned long long acc[16], prev;
    __device__ __forceinline__ void init() {
        for (int i = 0; i < 16; ++i) acc[i] = 0;
        prev = __builtin_amdgcn_s_memtime();
    }
    __device__ __forceinline__ void mark(int k) {
        const unsigned long long now = __builtin_amdgcn_s_memtime();
        acc[k] += now - prev;
        prev = now;
    }
    __device__ __forceinline__ void flush(int base) {
        if (threadIdx.x == 0)
            for (int i = 0; i < 16; ++i) atomicAdd(&g_mlp_stamps[base + i], acc[i]);
    }
};
#else
struct Stamps {
    __device__ __forceinline__ void init() {}
    __device__ __forceinline__ void mark(int) {}
    __device__ __forceinline__ void flush(int) {}
};
#endif

// shared forward: xB[T] -> h1B[2][T], h2B[2][T] (packed, relu'd; T 16-sample column tiles); weights from LDS fragments
template <int T>
__device__ __forceinline__ void forward_hidden(const __bf16 *frag, const float *sB1, const float *sB2, int lane,
                                               const bf16x8 xB[T], bf16x8 h1B[2][T], bf16x8 h2B[2][T], Stamps &st) {
    const int q = lane >> 4;
    f32x4 acc[4][T];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const bf16x8 w = ld_frag(frag, F_W1A + mt, lane);
        const f32x4 b = ld_bias4(sB1, 16 * mt + 4 * q);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[mt][t] = MFMA32(w, xB[t], b);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < T; ++t) h1B[s][t] = pack_relu(acc[2 * s][t], acc[2 * s + 1][t]);
    st.mark(1);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const f32x4 b = ld_bias4(sB2, 16 * mt + 4 * q);
#pragma unroll
        for (int t = 0; t < T; ++t) acc[mt][t] = b;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const bf16x8 w = ld_frag(frag, F_W2A + 2 * mt + s, lane);
#pragma unroll
            for (int t = 0; t < T; ++t) acc[mt][t] = MFMA32(w, h1B[s][t], acc[mt][t]);
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < T; ++t) h2B[s][t] = pack_relu(acc[2 * s][t], acc[2 * s + 1][t]);
    st.mark(2);
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
    bf16x8 xB[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int64_t m = (int64_t)blockIdx.x * 128 + w * 32 + 16 * t + c;
        xB[t] = load_x(a, clamp_row(m, M), q);
    }
    Stamps st;
    st.init();
    for (int64_t tile = blockIdx.x; tile * 128 < M; tile += gridDim.x) {
        const int64_t m0 = tile * 128 + w * 32;
        // the next tile's features are requested now and waited for after this tile's arithmetic
        bf16x8 xB_n[2], h1B[2][2], h2B[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int64_t m = m0 + (int64_t)gridDim.x * 128 + 16 * t + c;
            xB_n[t] = load_x(a, clamp_row(m, M), q);
        }
        st.mark(0);
        forward_hidden<2>(frag, sB1, sB2, lane, xB, h1B, h2B, st);
        const f32x4 b3 = ld_bias4(sB3, 4 * q);
        // (the loop-invariant fragments of layers 1 and 2 live in registers; layer 3's two are re-read from LDS every
        // tile through an index the optimiser cannot see through: hoisted as well they cost the third wave per SIMD)
        int lane3 = lane;
        asm("" : "+v"(lane3) : "s"((int)tile));   // no side effects (the other fragment reads stay hoisted), varies per tile
        const bf16x8 w3a = ld_frag(frag, F_W3A, lane3), w3b = ld_frag(frag, F_W3A + 1, lane3);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x4 o = b3;
            o = MFMA32(w3a, h2B[0][t], o);
            o = MFMA32(w3b, h2B[1][t], o);
            const int64_t m = m0 + 16 * t + c;  // lane holds h[4q + r] of sample m
            if (a.out_dim == 5) {
                // sigma + four latent channels: the q = 0 lane of a sample collects channel 4 from its q = 1 lane and
                // writes ONE 16-byte row (16 lanes = 256 contiguous bytes) instead of four scattered dwords -- the
                // store tail is where this kernel spends its time (tools/mlp_stamps.py)
                const float r3 = __shfl_down(o[0], 16, 64);
                if (q == 0 && m < M) {
                    sigmas[m] = expf(o[0] + blob_of(a, m));
                    reinterpret_cast<float4 *>(rgbs)[m] = make_float4(o[1], o[2], o[3], r3);
                }
            } else if (m < M) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = 4 * q + r;
                    if (n == 0) sigmas[m] = expf(o[r] + blob_of(a, m));
                    else if (n < a.out_dim) rgbs[m * nrgb + (n - 1)] = o[r];
                }
            }
        }
        st.mark(3);
        xB[0] = xB_n[0];
        xB[1] = xB_n[1];
    }
    st.flush(0);
}

// upstream gradient of one lane (outputs 4q .. 4q+3 of its T samples), requested one step ahead of its use: raw
// loads only -- d(sigma)/d(pre-activation) = sigma is applied when the values are consumed, so nothing waits here
template <int T>
struct Upstream { float v[T][4], sg[T]; };

template <int T>
__device__ __forceinline__ Upstream<T> load_upstream(const MlpArgs &a, const float *__restrict__ sigmas,
                                                     const float *__restrict__ dsigmas, const float *__restrict__ drgbs,
                                                     int64_t m0, int64_t M, int q, int c) {
    const int nrgb = a.out_dim - 1;
    Upstream<T> u;
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t m = m0 + 16 * t + c;
        const bool in = m < M;
        const int64_t mc = clamp_row(m, M);
        if (a.out_dim == 5) {
            // sigma + four latent channels: every lane reads its sample's dsigma, sigma and the 16-byte row of latent
            // gradients (the four q groups of a sample read the same addresses: one access), then keeps its share --
            // no exec-masked branches around the loads, three loads per tile
            const float ds = dsigmas[mc], sg = sigmas[mc];
            const float4 g = reinterpret_cast<const float4 *>(drgbs)[mc];
            const bool q0 = in && q == 0, q1 = in && q == 1;
            u.sg[t] = sg;
            u.v[t][0] = q0 ? ds : (q1 ? g.w : 0.f);
            u.v[t][1] = q0 ? g.x : 0.f;
            u.v[t][2] = q0 ? g.y : 0.f;
            u.v[t][3] = q0 ? g.z : 0.f;
        } else {
            u.sg[t] = 1.0f;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) u.v[t][jj] = 0.f;
            if (in && q == 0) { u.v[t][0] = dsigmas[m]; u.sg[t] = sigmas[m]; }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                const int n = 4 * q + jj;
                if (in && n >= 1 && n < a.out_dim) u.v[t][jj] = drgbs[m * nrgb + (n - 1)];
            }
        }
    }
    return u;
}
// the B fragment of dZ3^T (slot (q, jj < 4) <-> output 4q + jj) of one column tile; returns whether any value is non-zero
__device__ __forceinline__ bool upstream_fragment(const float v[4], float sg, int q, bf16x8 &d3) {
    const float e15 = 3269017.3724721107f;  // exp(15)
    const float v0 = q == 0 ? v[0] * fminf(sg, e15) : v[0];
    Pk8 d;
    d.u[0] = cvt_pk(v0, v[1]);
    d.u[1] = cvt_pk(v[2], v[3]);
    d.u[2] = 0u;
    d.u[3] = 0u;
    d3 = d.v;
    return (v0 != 0.f) || (v[1] != 0.f) || (v[2] != 0.f) || (v[3] != 0.f);
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

// NW wavefronts x T 16-sample column tiles per wavefront = one step of NW * 16 * T samples.  <4, 2> (two workgroups =
// eight wavefronts per CU) is what runs; <8, 1> (half the registers per wavefront, four wavefronts per SIMD) measured
// 65 us against 60: the kernel is not short of wavefronts but of LDS bandwidth and issue slots.
// Ownership of the weight gradients (r = w & 3, h = w >> 2, NH = NW / 4): rows 16r..16r+15 of dW2 / dW1, their column
// tiles split over h; db2 with h = 0, db1 with h = NH - 1; dW3 columns 16r.. with h = 0; db3 with the first wave of
// the last h.  No cross-wave reduction.
template <int NW, int T>
__global__ void __launch_bounds__(NW * 64, (NW == 4 ? 2 : 4))
k_mlp_backward_bf16(MlpArgs a, const float *__restrict__ sigmas, const float *__restrict__ dsigmas,
                    const float *__restrict__ drgbs, float *__restrict__ dfeat, float *__restrict__ slabs) {
    constexpr int STEP = NW * 16 * T, KS = STEP / 32, NH = NW / 4, C2 = 4 / NH, C1 = 2 / NH;
    __shared__ __attribute__((aligned(16))) __bf16 frag[F_ALL * 512];
    __shared__ __attribute__((aligned(16))) __bf16 imgA[STEP * RS];  // [sample][feature]: H2, then H1, then X
    __shared__ __attribute__((aligned(16))) __bf16 imgD[STEP * RS];  // [sample][feature]: dZ3, then dZ2, then dZ1
    __shared__ float sB1[MLP_HID], sB2[MLP_HID];
    int64_t M = a.m_host;
    if (a.m_dev) { const int64_t md = *a.m_dev; M = md < M ? md : M; }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, c = lane & 15;
    const int r = w & 3, h = w >> 2;
    fetch_fragments(a, frag, F_ALL, tid, NW * 64);
    if (tid < MLP_HID) { sB1[tid] = a.b1[tid]; sB2[tid] = a.b2[tid]; }
    __syncthreads();
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;

    f32x4 gW2[C2], gW1[C1], gB = zero4, gW3 = zero4;   // gB: db2 (h = 0) or db1 (h = NH-1); NW = 4 keeps both
    f32x4 gB1x = zero4, gB3 = zero4;
#pragma unroll
    for (int i = 0; i < C2; ++i) gW2[i] = zero4;
#pragma unroll
    for (int i = 0; i < C1; ++i) gW1[i] = zero4;
    const bool own_b2 = h == 0, own_b1 = h == NH - 1, own_w3 = h == 0, own_b3 = w == NW - 4;

    // this wave's inputs of the first step; every later step's are requested one step ahead
    const int64_t wofs = (int64_t)w * 16 * T;
    Upstream<T> up = load_upstream<T>(a, sigmas, dsigmas, drgbs, (int64_t)blockIdx.x * STEP + wofs, M, q, c);
    bf16x8 xB[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int64_t m = (int64_t)blockIdx.x * STEP + wofs + 16 * t + c;
        xB[t] = load_x(a, clamp_row(m, M), q);
    }
    Stamps st;
    st.init();
    for (int64_t tile = blockIdx.x; tile * STEP < M; tile += gridDim.x) {
        const int64_t m0 = tile * STEP + wofs, m1 = m0 + (int64_t)gridDim.x * STEP;
        const Upstream<T> up_c = up;
        bf16x8 xC[T], h1B[2][T], h2B[2][T], dzB[2][T], d3B[T];
#pragma unroll
        for (int t = 0; t < T; ++t) xC[t] = xB[t];
        up = load_upstream<T>(a, sigmas, dsigmas, drgbs, m1, M, q, c);
#pragma unroll
        for (int t = 0; t < T; ++t) xB[t] = load_x(a, clamp_row(m1 + 16 * t + c, M), q);
        bool live = false;
#pragma unroll
        for (int t = 0; t < T; ++t) live = upstream_fragment(up_c.v[t], up_c.sg[t], q, d3B[t]) || live;
        // (A step whose upstream gradient is exactly zero -- rays past their termination point, 8 % of the bench's
        // samples -- used to be skipped behind a __syncthreads_or.  The skip made every accumulator live across a branch:
        // ~130 register copies per step at the merge, more than the skipped arithmetic was worth; a dead step now flows
        // through and produces its zeros: dfeat = +0, nothing added to any weight gradient.)
        (void)live;
        st.mark(4);   // __syncthreads_or
        forward_hidden<T>(frag, sB1, sB2, lane, xC, h1B, h2B, st);   // marks 1, 2
        // ================= stage 1: dW3 += dZ3^T (x) H2^T
        __syncthreads();  // previous step's readers of imgA/imgD are done
        st.mark(5);
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int row = 16 * T * w + 16 * t + c;  // this lane's sample inside the step
#pragma unroll
            for (int s = 0; s < 2; ++s) stage_pair(imgA, row, 32 * s + 4 * q, 32 * s + 16 + 4 * q, h2B[s][t]);
            stage_lo(imgD, row, 4 * q, d3B[t]);   // outputs 4q .. 4q+3 (columns 0..15; zero beyond out_dim)
        }
        __syncthreads();
        st.mark(6);   // stage-1 images written + barrier
        if (own_w3 || own_b3) {   // wave-uniform
#pragma unroll
            for (int k = 0; k < KS; ++k) {
                const bf16x8 dA = ld_tr(imgD, k, 0, lane);        // A[i = output][k = sample]
                if (own_w3) gW3 = MFMA32(dA, ld_tr(imgA, k, 16 * r, lane), gW3);   // B[k = sample][j = hidden 16r + c]
                if (own_b3) gB3 = MFMA32(dA, ones, gB3);
            }
        }
        st.mark(7);   // dW3
        // ---- dA2 = W3^T dZ3 ; dZ2 = dA2 masked by H2 > 0
        {
            f32x4 acc[4][T];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const bf16x8 wf = ld_frag(frag, F_W3T + mt, lane);
#pragma unroll
                for (int t = 0; t < T; ++t) acc[mt][t] = MFMA32(wf, d3B[t], zero4);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < T; ++t) dzB[s][t] = pack_masked(acc[2 * s][t], acc[2 * s + 1][t], h2B[s][t]);
        }
        st.mark(8);   // dA2 chain
        // ================= stage 2: dW2 += dZ2^T (x) H1^T
        __syncthreads();
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int row = 16 * T * w + 16 * t + c;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                stage_pair(imgA, row, 32 * s + 4 * q, 32 * s + 16 + 4 * q, h1B[s][t]);
                stage_pair(imgD, row, 32 * s + 4 * q, 32 * s + 16 + 4 * q, dzB[s][t]);
            }
        }
        __syncthreads();
        st.mark(9);   // stage-2 barrier + images + barrier
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const bf16x8 dA = ld_tr(imgD, k, 16 * r, lane);
#pragma unroll
            for (int i = 0; i < C2; ++i) gW2[i] = MFMA32(dA, ld_tr(imgA, k, 16 * (C2 * h + i), lane), gW2[i]);
            if (own_b2) gB = MFMA32(dA, ones, gB);
        }
        st.mark(10);  // dW2
        // ---- dA1 = W2^T dZ2 ; dZ1 = dA1 masked by H1 > 0   (dzB is overwritten by dZ1)
        {
            f32x4 acc[4][T];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int t = 0; t < T; ++t) acc[mt][t] = zero4;
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const bf16x8 wf = ld_frag(frag, F_W2T + 2 * mt + s, lane);
#pragma unroll
                    for (int t = 0; t < T; ++t) acc[mt][t] = MFMA32(wf, dzB[s][t], acc[mt][t]);
                }
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int t = 0; t < T; ++t) dzB[s][t] = pack_masked(acc[2 * s][t], acc[2 * s + 1][t], h1B[s][t]);
        }
        st.mark(11);  // dA1 chain
        // ================= stage 3: dW1 += dZ1^T (x) X^T
        __syncthreads();
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int row = 16 * T * w + 16 * t + c;
            stage_pair(imgA, row, 8 * q, 8 * q + 4, xC[t]);   // input features 8q .. 8q+7
#pragma unroll
            for (int s = 0; s < 2; ++s) stage_pair(imgD, row, 32 * s + 4 * q, 32 * s + 16 + 4 * q, dzB[s][t]);
        }
        __syncthreads();
        st.mark(12);  // stage-3 barrier + images + barrier
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            const bf16x8 dA = ld_tr(imgD, k, 16 * r, lane);
#pragma unroll
            for (int i = 0; i < C1; ++i) gW1[i] = MFMA32(dA, ld_tr(imgA, k, 16 * (C1 * h + i), lane), gW1[i]);
            if (own_b1) { if (NH == 1) gB1x = MFMA32(dA, ones, gB1x); else gB = MFMA32(dA, ones, gB); }
        }
        st.mark(13);  // dW1
        // ---- dX = W1^T dZ1 -> dfeat (level-major f32): lane holds features 16mt + 4q + r of its sample
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f32x4 ax[T];
#pragma unroll
            for (int t = 0; t < T; ++t) ax[t] = zero4;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const bf16x8 wf = ld_frag(frag, F_W1T + 2 * mt + s, lane);
#pragma unroll
                for (int t = 0; t < T; ++t) ax[t] = MFMA32(wf, dzB[s][t], ax[t]);
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int64_t m = m0 + 16 * t + c;
                if (m < M) {
                    const int lv = 8 * mt + 2 * q;  // features 16mt+4q+{0,1} = level lv, {2,3} = level lv+1
                    char *df = reinterpret_cast<char *>(dfeat);
                    const uint32_t ls = (uint32_t)a.level_stride, o0 = (((uint32_t)lv) * ls + (uint32_t)m) << 3;
                    *reinterpret_cast<float2 *>(df + (size_t)o0) = make_float2(ax[t][0], ax[t][1]);
                    *reinterpret_cast<float2 *>(df + (size_t)(o0 + (ls << 3))) = make_float2(ax[t][2], ax[t][3]);
                }
            }
        }
        st.mark(14);  // dX + stores issued
    }

    st.flush(16);
    // ---- one slab per workgroup, every wave writes the tiles it owns (layout: mlp_shared.h)
    float *slab = slabs + (int64_t)blockIdx.x * MLP_SLAB;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
        const int o = 16 * r + 4 * q + rr;  // row of dW2 / dW1
#pragma unroll
        for (int i = 0; i < C2; ++i) slab[MLP_SL_W2 + o * MLP_HID + 16 * (C2 * h + i) + c] = gW2[i][rr];
#pragma unroll
        for (int i = 0; i < C1; ++i) slab[MLP_SL_W1 + o * MLP_IN + 16 * (C1 * h + i) + c] = gW1[i][rr];
        if (c == 0) {
            if (own_b2) slab[MLP_SL_B2 + o] = gB[rr];
            if (own_b1) slab[MLP_SL_B1 + o] = NH == 1 ? gB1x[rr] : gB[rr];
        }
        // dW3: rows n3 = 4q + rr, columns (hidden) 16r + c
        if (own_w3) slab[MLP_SL_W3 + (4 * q + rr) * MLP_HID + 16 * r + c] = gW3[rr];
        if (own_b3 && c == 0) slab[MLP_SL_B3 + 4 * q + rr] = gB3[rr];
    }
}

// ------------------------------------------------------------------ backward, operand-swap form
// The weight gradients contract over SAMPLES, so their MFMA operands need the feature on the lane and samples in the
// registers -- the transpose of what the sample-on-the-lane chain holds.  Instead of transposing through LDS, every
// activation / pre-activation gradient the weight gradients need is computed a second time with the two MFMA operands
// SWAPPED: A[i][k] and B[k][j] have the same lane map (index on the lane, k in the registers), so
// MFMA(W-fragment, X^T-fragment) = Z^T (feature rows, sample columns: the chain) and MFMA(X^T-fragment, W-fragment) = Z
// (sample rows 4q+r in the registers, feature column c on the lane) use the very same registers.  Two 16-sample
// tiles of Z give 8 samples per lane = one k-step of dW = dZ^T (x) H (the k-slot -> sample map is the same for both
// operands, which is all a contraction needs).  +54 MFMAs per 32 samples (a few us chip-wide), and in exchange: no
// staging images, no transposing reads, no LDS round trips and NO BARRIER in the loop -- a wave carries its own 32
// samples from load to store; the next step's inputs are requested before the current step is computed.
// X and dZ3 (inputs, no weight to swap with) are transposed exactly by MFMAs against 0/1 selection fragments.
constexpr int F_SELX = 30;   // [nt 0..1]  B[k = 8q+jj][j = c] = (8q + jj == 16nt + c)
constexpr int F_SEL3 = 32;   //            B[k = 8q+jj][j = c] = (jj < 4 && 4q + jj == c)
constexpr int F_SW = 33;
constexpr int SW_TILES = 31;                                     // dW2 16 | dW1 8 | dW3 4 | biases 3 (db2 4, db1 4, db3 1 floats)
static_assert((size_t)F_SW * 1024 <= MLP_FRAG_BYTES, "fragment cache");
static_assert(SW_TILES * 1024 <= F_SW * 1024, "final reduction reuses the fragment area");

// the three selection fragments (appended to the weight fragments by k_mlp_build_fragments)
__device__ __forceinline__ void build_selectors(__bf16 *frag, int tid, int nthreads) {
    for (int e = tid; e < (F_SW - F_SELX) * 512; e += nthreads) {
        const int f = F_SELX + (e >> 9), l = (e >> 3) & 63, jj = e & 7;
        const int q = l >> 4, c = l & 15;
        const bool one = f < F_SEL3 ? (8 * q + jj == 16 * (f - F_SELX) + c) : (jj < 4 && 4 * q + jj == c);
        frag[F_SELX * 512 + e] = (__bf16)(one ? 1.0f : 0.0f);
    }
}

// (operand-swap form of the backward: measured slower than the default at either occupancy, spills at two waves per
// SIMD -- DESIGN.md section 4 H7; compiled only into experiment builds: -DLNERF_EXPERIMENTS)
#ifdef LNERF_EXPERIMENTS
template <int WPS>
__global__ void __launch_bounds__(256, WPS)
k_mlp_backward_bf16_sw(MlpArgs a, const float *__restrict__ sigmas, const float *__restrict__ dsigmas,
                       const float *__restrict__ drgbs, float *__restrict__ dfeat, float *__restrict__ slabs) {
    __shared__ __attribute__((aligned(16))) __bf16 frag[F_SW * 512];
    __shared__ float sB1[MLP_HID], sB2[MLP_HID];
    int64_t M = a.m_host;
    if (a.m_dev) { const int64_t md = *a.m_dev; M = md < M ? md : M; }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, q = lane >> 4, c = lane & 15;
    fetch_fragments(a, frag, F_SW, tid, 256);
    if (tid < MLP_HID) { sB1[tid] = a.b1[tid]; sB2[tid] = a.b2[tid]; }
    __syncthreads();
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 gW2[16], gW1[8], gW3[4];
    float gb2[4], gb1[4], gb3 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) gW2[i] = zero4;
#pragma unroll
    for (int i = 0; i < 8; ++i) gW1[i] = zero4;
#pragma unroll
    for (int i = 0; i < 4; ++i) { gW3[i] = zero4; gb2[i] = 0.f; gb1[i] = 0.f; }

    const int64_t stride = (int64_t)gridDim.x * 4 * 32;
    int64_t m0 = ((int64_t)blockIdx.x * 4 + w) * 32;
    Upstream<2> up = load_upstream<2>(a, sigmas, dsigmas, drgbs, m0, M, q, c);
    bf16x8 xB[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) xB[t] = load_x(a, clamp_row(m0 + 16 * t + c, M), q);

    while (m0 < M) {   // wave-uniform
        asm volatile("" ::: "memory");   // the weight fragments are re-read from LDS every step, not hoisted into ~130 registers
        const int64_t m1 = m0 + stride;
        const Upstream<2> up_n = load_upstream<2>(a, sigmas, dsigmas, drgbs, m1, M, q, c);
        bf16x8 xB_n[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) xB_n[t] = load_x(a, clamp_row(m1 + 16 * t + c, M), q);

        bf16x8 d3B[2];
        bool live = false;
#pragma unroll
        for (int t = 0; t < 2; ++t) live = upstream_fragment(up.v[t], up.sg[t], q, d3B[t]) || live;
        if (!__any(live ? 1 : 0)) {
            // 32 samples whose upstream gradient is exactly zero (rays past their termination point: the compositing
            // backward writes zeros there) contribute nothing to any gradient: dfeat = 0, done
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int64_t m = m0 + 16 * t + c;
                if (m < M) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        reinterpret_cast<float2 *>(dfeat)[(int64_t)(4 * q + k) * a.level_stride + m] = make_float2(0.f, 0.f);
                }
            }
        } else {
            bf16x8 h1B[2][2], h2B[2][2], H1f[4], H2f[4], dZf[4];
            // ---- layer 1, both forms (the fragment of W1 rows 16mt.. is A of the chain and B of the swap)
            {
                f32x4 acc[4][2];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const bf16x8 wf = ld_frag(frag, F_W1A + mt, lane);
                    const f32x4 b = ld_bias4(sB1, 16 * mt + 4 * q);
                    acc[mt][0] = MFMA32(wf, xB[0], b);
                    acc[mt][1] = MFMA32(wf, xB[1], b);
                    const float bc = sB1[16 * mt + c];
                    const f32x4 bT = (f32x4){bc, bc, bc, bc};
                    H1f[mt] = pack_relu_pk(MFMA32(xB[0], wf, bT), MFMA32(xB[1], wf, bT));
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int t = 0; t < 2; ++t) h1B[s][t] = pack_relu_pk(acc[2 * s][t], acc[2 * s + 1][t]);
            }
            // ---- layer 2, both forms
            {
                f32x4 acc[4][2];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const f32x4 b = ld_bias4(sB2, 16 * mt + 4 * q);
                    const float bc = sB2[16 * mt + c];
                    f32x4 t0 = (f32x4){bc, bc, bc, bc}, t1 = t0;
                    acc[mt][0] = b;
                    acc[mt][1] = b;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 wf = ld_frag(frag, F_W2A + 2 * mt + s, lane);
                        acc[mt][0] = MFMA32(wf, h1B[s][0], acc[mt][0]);
                        acc[mt][1] = MFMA32(wf, h1B[s][1], acc[mt][1]);
                        t0 = MFMA32(h1B[s][0], wf, t0);
                        t1 = MFMA32(h1B[s][1], wf, t1);
                    }
                    H2f[mt] = pack_relu_pk(t0, t1);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int t = 0; t < 2; ++t) h2B[s][t] = pack_relu_pk(acc[2 * s][t], acc[2 * s + 1][t]);
            }
            // ---- dW3 += dZ3^T (x) H2 ; db3
            {
                const bf16x8 sel = ld_frag(frag, F_SEL3, lane);
                const bf16x8 d3f = pack_plain_pk(MFMA32(d3B[0], sel, zero4), MFMA32(d3B[1], sel, zero4));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) gW3[nt] = MFMA32(d3f, H2f[nt], gW3[nt]);
                gb3 = sum8(d3f, gb3);
            }
            // ---- dA2 = dZ3 W3, both forms ; dZ2 = dA2 masked by H2 > 0 ; dW2 += dZ2^T (x) H1 ; db2
            bf16x8 dzB[2][2];
            {
                f32x4 acc[4][2];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const bf16x8 wf = ld_frag(frag, F_W3T + mt, lane);
                    acc[mt][0] = MFMA32(wf, d3B[0], zero4);
                    acc[mt][1] = MFMA32(wf, d3B[1], zero4);
                    dZf[mt] = pack_masked_pk(MFMA32(d3B[0], wf, zero4), MFMA32(d3B[1], wf, zero4), H2f[mt]);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int t = 0; t < 2; ++t) dzB[s][t] = pack_masked_pk(acc[2 * s][t], acc[2 * s + 1][t], h2B[s][t]);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) gW2[4 * mt + nt] = MFMA32(dZf[mt], H1f[nt], gW2[4 * mt + nt]);
                gb2[mt] = sum8(dZf[mt], gb2[mt]);
            }
            // ---- dA1 = dZ2 W2, both forms ; dZ1 = dA1 masked by H1 > 0   (dzB is overwritten by dZ1)
            {
                f32x4 acc[4][2];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    f32x4 t0 = zero4, t1 = zero4;
                    acc[mt][0] = acc[mt][1] = zero4;
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const bf16x8 wf = ld_frag(frag, F_W2T + 2 * mt + s, lane);
                        acc[mt][0] = MFMA32(wf, dzB[s][0], acc[mt][0]);
                        acc[mt][1] = MFMA32(wf, dzB[s][1], acc[mt][1]);
                        t0 = MFMA32(dzB[s][0], wf, t0);
                        t1 = MFMA32(dzB[s][1], wf, t1);
                    }
                    dZf[mt] = pack_masked_pk(t0, t1, H1f[mt]);
                }
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int t = 0; t < 2; ++t) dzB[s][t] = pack_masked_pk(acc[2 * s][t], acc[2 * s + 1][t], h1B[s][t]);
            }
            // ---- dW1 += dZ1^T (x) X ; db1
            {
                bf16x8 Xf[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const bf16x8 sel = ld_frag(frag, F_SELX + nt, lane);
                    Xf[nt] = pack_plain_pk(MFMA32(xB[0], sel, zero4), MFMA32(xB[1], sel, zero4));
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) gW1[2 * mt + nt] = MFMA32(dZf[mt], Xf[nt], gW1[2 * mt + nt]);
                    gb1[mt] = sum8(dZf[mt], gb1[mt]);
                }
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
        up = up_n;
        xB[0] = xB_n[0];
        xB[1] = xB_n[1];
        m0 = m1;
    }

    // ---- bias partials: a lane holds feature c of the samples of its q group -> sum the four groups
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        gb2[i] += __shfl_xor(gb2[i], 16);
        gb2[i] += __shfl_xor(gb2[i], 32);
        gb1[i] += __shfl_xor(gb1[i], 16);
        gb1[i] += __shfl_xor(gb1[i], 32);
    }
    gb3 += __shfl_xor(gb3, 16);
    gb3 += __shfl_xor(gb3, 32);
    f32x4 gB[3] = {(f32x4){gb2[0], gb2[1], gb2[2], gb2[3]}, (f32x4){gb1[0], gb1[1], gb1[2], gb1[3]},
                   (f32x4){gb3, 0.f, 0.f, 0.f}};
    // ---- sum the four waves' tiles in the fixed order 0 + 1 + 2 + 3 (through the fragment area, now idle)
    f32x4 *red = reinterpret_cast<f32x4 *>(frag);
    for (int src = 1; src < 4; ++src) {
        __syncthreads();
        if (w == src) {
#pragma unroll
            for (int i = 0; i < 16; ++i) red[i * 64 + lane] = gW2[i];
#pragma unroll
            for (int i = 0; i < 8; ++i) red[(16 + i) * 64 + lane] = gW1[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) red[(24 + i) * 64 + lane] = gW3[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) red[(28 + i) * 64 + lane] = gB[i];
        }
        __syncthreads();
        if (w == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) gW2[i] += red[i * 64 + lane];
#pragma unroll
            for (int i = 0; i < 8; ++i) gW1[i] += red[(16 + i) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) gW3[i] += red[(24 + i) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 3; ++i) gB[i] += red[(28 + i) * 64 + lane];
        }
    }
    if (w != 0) return;
    float *slab = slabs + (int64_t)blockIdx.x * MLP_SLAB;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int o = 16 * mt + 4 * q + r;  // row of dW2 / dW1
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) slab[MLP_SL_W2 + o * MLP_HID + 16 * nt + c] = gW2[4 * mt + nt][r];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) slab[MLP_SL_W1 + o * MLP_IN + 16 * nt + c] = gW1[2 * mt + nt][r];
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) slab[MLP_SL_W3 + (4 * q + r) * MLP_HID + 16 * nt + c] = gW3[nt][r];
    }
    // biases: lane c (of q group 0) holds db[16 mt + c]
    if (q == 0) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            slab[MLP_SL_B2 + 16 * mt + c] = gB[0][mt];
            slab[MLP_SL_B1 + 16 * mt + c] = gB[1][mt];
        }
        slab[MLP_SL_B3 + c] = gB[2][0];
    }
}
#endif  // LNERF_EXPERIMENTS

}  // namespace lnerf

namespace lnerf {

#ifdef LNERF_STAMPS
int mlp_stamps_read(unsigned long long *out32) {
    unsigned long long z[32] = {0};
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_mlp_stamps), sizeof(z)) != hipSuccess) return LNERF_ERR_HIP;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_mlp_stamps), z, sizeof(z)) != hipSuccess) return LNERF_ERR_HIP;
    return LNERF_OK;
}
#endif

int launch_mlp_fragment_maps(int out_dim, int32_t *m1, int32_t *m2, int32_t *m3, hipStream_t stream) {
    if (hipMemsetAsync(m1, 0xFF, (size_t)MLP_HID * MLP_IN * 2 * sizeof(int32_t), stream) != hipSuccess ||
        hipMemsetAsync(m2, 0xFF, (size_t)MLP_HID * MLP_HID * 2 * sizeof(int32_t), stream) != hipSuccess ||
        hipMemsetAsync(m3, 0xFF, (size_t)out_dim * MLP_HID * 2 * sizeof(int32_t), stream) != hipSuccess) {
        set_error("mlp_fragment_maps: hipMemsetAsync failed");
        return LNERF_ERR_HIP;
    }
    hipLaunchKernelGGL(k_mlp_fragment_maps, dim3(16), dim3(256), 0, stream, out_dim, m1, m2, m3);
    LNERF_CHECK_LAUNCH("mlp(fragment maps)");
    return LNERF_OK;
}

int launch_mlp_fragments_bf16(const MlpArgs &a, void *frag_out, bool backward_too, hipStream_t stream) {
    const int n = backward_too ? F_SW : F_FWD;
    hipLaunchKernelGGL(k_mlp_build_fragments, dim3((unsigned)(n * 2)), dim3(256), 0, stream, a, (__bf16 *)frag_out, n);
    LNERF_CHECK_LAUNCH("mlp(fragments)");
    return LNERF_OK;
}

int launch_mlp_forward_bf16(const MlpArgs &a, float *sigmas, float *rgbs, int blocks, int wps, hipStream_t stream) {
#ifdef LNERF_EXPERIMENTS   // (four waves per SIMD: 128 registers, 396 bytes of scratch per lane -- measured slower)
    if (wps >= 4) hipLaunchKernelGGL(k_mlp_forward_bf16<4>, dim3((unsigned)blocks), dim3(256), 0, stream, a, sigmas, rgbs);
    else
#endif
    if (wps >= 3) hipLaunchKernelGGL(k_mlp_forward_bf16<3>, dim3((unsigned)blocks), dim3(256), 0, stream, a, sigmas, rgbs);
    else hipLaunchKernelGGL(k_mlp_forward_bf16<2>, dim3((unsigned)blocks), dim3(256), 0, stream, a, sigmas, rgbs);
    LNERF_CHECK_LAUNCH("mlp_forward(bf16)");
    return LNERF_OK;
}

int launch_mlp_backward_bf16(const MlpArgs &a, const float *sigmas, const float *dsigmas, const float *drgbs,
                             float *dfeat, float *slabs, int blocks, int variant, hipStream_t stream) {
#ifdef LNERF_EXPERIMENTS
    if (variant == 2) {
        hipLaunchKernelGGL(k_mlp_backward_bf16_sw<2>, dim3((unsigned)blocks), dim3(256), 0, stream, a, sigmas, dsigmas,
                           drgbs, dfeat, slabs);
        LNERF_CHECK_LAUNCH("mlp_backward(bf16, operand swap)");
        return LNERF_OK;
    }
    if (variant == 1) {
        hipLaunchKernelGGL(k_mlp_backward_bf16_sw<1>, dim3((unsigned)blocks), dim3(256), 0, stream, a, sigmas, dsigmas,
                           drgbs, dfeat, slabs);
        LNERF_CHECK_LAUNCH("mlp_backward(bf16, operand swap)");
        return LNERF_OK;
    }
#endif
    (void)variant;
    hipLaunchKernelGGL((k_mlp_backward_bf16<4, 2>), dim3((unsigned)blocks), dim3(256), 0, stream, a, sigmas, dsigmas,
                       drgbs, dfeat, slabs);
    LNERF_CHECK_LAUNCH("mlp_backward(bf16)");
    return LNERF_OK;
}

}  // namespace lnerf
