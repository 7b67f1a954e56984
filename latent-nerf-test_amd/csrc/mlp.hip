// H7 fused sigma/latent MLP  32 -> 64 -> 64 -> out_dim, forward and backward, on the matrix cores.
//
// precision LNERF_F32: v_mfma_f32_16x16x4_f32 (exact f32 fma chains, the parity path).
// One wavefront owns 16 samples per step; a 256-thread workgroup owns a 64-sample tile and
// walks tiles persistently.  Activations change from the MFMA C/D layout (sample on
// registers, feature on lanes) to the A layout (sample on lanes) through a per-wave LDS tile.
//
// MFMA 16x16x4 f32 lane maps (guide §3):  A[i = l&15][k = l>>4],  B[k = l>>4][j = l&15],
// C/D[i = (l>>4)*4 + reg][j = l&15].
#include "common.h"
#include "mlp_shared.h"

namespace lnerf {

#define MFMA4(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int IN = MLP_IN, HID = MLP_HID, OUTP = MLP_OUTP;
constexpr int LDX = IN + 1, LDH = HID + 1;   // padded LDS leading dimensions

__device__ __forceinline__ float load_feat(const MlpArgs &a, int level, int f, int64_t m) {
    const int64_t i = ((int64_t)level * a.level_stride + m) * 2 + f;
    if (a.feat_bf16) return bf16_to_f32(reinterpret_cast<const uint16_t *>(a.feat)[i]);
    return reinterpret_cast<const float *>(a.feat)[i];
}

// ------------------------------------------------------------------ forward
__global__ void __launch_bounds__(256)
k_mlp_forward_f32(MlpArgs a, float *__restrict__ sigmas, float *__restrict__ rgbs) {
    __shared__ float sAct[4][16 * LDH];
    int64_t M = a.m_host;
    if (a.m_dev) { const int64_t md = *a.m_dev; M = md < M ? md : M; }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    float *act = sAct[w];

    // weights as B operands, held in registers for the whole kernel: B[k][n] = W[n][k]
    float w1r[8][4], w2r[16][4], w3r[16];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) w1r[kk][nt] = a.w1[(nt * 16 + j) * IN + 4 * kk + q];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) w2r[kk][nt] = a.w2[(nt * 16 + j) * HID + 4 * kk + q];
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) w3r[kk] = (j < a.out_dim) ? a.w3[j * HID + 4 * kk + q] : 0.f;
    float b1r[4], b2r[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { b1r[nt] = a.b1[nt * 16 + j]; b2r[nt] = a.b2[nt * 16 + j]; }
    const float b3r = (j < a.out_dim) ? a.b3[j] : 0.f;

    for (int64_t tile = blockIdx.x; tile * 64 < M; tile += gridDim.x) {
        const int64_t m0 = tile * 64 + w * 16;
        // ---- layer 1: A[s][k] straight from the level-major feature tensor
        f32x4 acc[4];
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){b1r[nt], b1r[nt], b1r[nt], b1r[nt]};
        {
            const int64_t m = m0 + j;
            const bool in = m < M;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float x = in ? load_feat(a, 2 * kk + (q >> 1), q & 1, m) : 0.f;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA4(x, w1r[kk][nt], acc[nt]);
            }
        }
        __syncthreads();  // previous tile's readers of `act` are done
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) act[(q * 4 + r) * LDH + nt * 16 + j] = fmaxf(acc[nt][r], 0.f);
        __syncthreads();
        // ---- layer 2
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = (f32x4){b2r[nt], b2r[nt], b2r[nt], b2r[nt]};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float x = act[j * LDH + 4 * kk + q];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = MFMA4(x, w2r[kk][nt], acc[nt]);
        }
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) act[(q * 4 + r) * LDH + nt * 16 + j] = fmaxf(acc[nt][r], 0.f);
        __syncthreads();
        // ---- layer 3 (one 16-wide output tile, columns >= out_dim are zero)
        f32x4 o = (f32x4){b3r, b3r, b3r, b3r};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) o = MFMA4(act[j * LDH + 4 * kk + q], w3r[kk], o);
        // ---- epilogue: lane holds h[s = q*4 + r][n = j]
        if (j < a.out_dim) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = m0 + q * 4 + r;
                if (m < M) {
                    if (j == 0) sigmas[m] = expf(o[r] + blob_of(a, m));
                    else rgbs[m * (a.out_dim - 1) + (j - 1)] = o[r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------ backward
constexpr int SL_W1 = MLP_SL_W1, SL_B1 = MLP_SL_B1, SL_W2 = MLP_SL_W2, SL_B2 = MLP_SL_B2, SL_W3 = MLP_SL_W3,
              SL_B3 = MLP_SL_B3, SLAB = MLP_SLAB;
constexpr int BWD_MAX_BLOCKS = MLP_BWD_MAX_BLOCKS;

__global__ void __launch_bounds__(256, 2)
k_mlp_backward_f32(MlpArgs a, const float *__restrict__ sigmas, const float *__restrict__ dsigmas,
                   const float *__restrict__ drgbs, float *__restrict__ dfeat, float *__restrict__ slabs) {
    // one LDS block, carved by hand so that the per-wave activation tiles can be reused for the
    // end-of-kernel reduction of the four waves' weight-gradient accumulators
    constexpr int O_W1 = 0, O_W2 = O_W1 + HID * LDX, O_W3 = O_W2 + HID * LDH, O_B1 = O_W3 + OUTP * LDH,
                  O_B2 = O_B1 + HID, O_X = O_B2 + HID, O_P = O_X + 4 * 16 * LDX, O_Q = O_P + 4 * 16 * LDH,
                  O_END = O_Q + 4 * 16 * LDH;
    static_assert(O_END - O_X >= SLAB, "activation tiles must hold one slab");
    __shared__ float smem[O_END];
    float *sW1 = smem + O_W1, *sW2 = smem + O_W2, *sW3 = smem + O_W3, *sB1 = smem + O_B1, *sB2 = smem + O_B2;
    int64_t M = a.m_host;
    if (a.m_dev) { const int64_t md = *a.m_dev; M = md < M ? md : M; }
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int j = lane & 15, q = lane >> 4;
    const int nrgb = a.out_dim - 1;
    for (int i = tid; i < HID * IN; i += 256) sW1[(i / IN) * LDX + (i % IN)] = a.w1[i];
    for (int i = tid; i < HID * HID; i += 256) sW2[(i / HID) * LDH + (i % HID)] = a.w2[i];
    for (int i = tid; i < OUTP * HID; i += 256)
        sW3[(i / HID) * LDH + (i % HID)] = (i / HID) < a.out_dim ? a.w3[i] : 0.f;
    if (tid < HID) { sB1[tid] = a.b1[tid]; sB2[tid] = a.b2[tid]; }
    __syncthreads();
    float *X = smem + O_X + w * 16 * LDX, *P = smem + O_P + w * 16 * LDH, *Q = smem + O_Q + w * 16 * LDH;
    const float e15 = 3269017.3724721107f;  // exp(15)

    f32x4 gW2[4][4], gW1[4][2], gW3[4];
    float gb1[4] = {0.f, 0.f, 0.f, 0.f}, gb2[4] = {0.f, 0.f, 0.f, 0.f}, gb3[2] = {0.f, 0.f};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) gW2[mt][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gW1[mt][0] = gW1[mt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        gW3[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

    for (int64_t tile = blockIdx.x; tile * 64 < M; tile += gridDim.x) {
        const int64_t m0 = tile * 64 + w * 16;
        __syncthreads();  // previous tile's LDS readers are done
        // ---- recompute layer 1; keep X in LDS ([s][k]) for dW1
        f32x4 z[4], a1c[4], a2c[4];
        {
            const int64_t m = m0 + j;
            const bool in = m < M;
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { const float b = sB1[nt * 16 + j]; z[nt] = (f32x4){b, b, b, b}; }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const float x = in ? load_feat(a, 2 * kk + (q >> 1), q & 1, m) : 0.f;
                X[j * LDX + 4 * kk + q] = x;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) z[nt] = MFMA4(x, sW1[(nt * 16 + j) * LDX + 4 * kk + q], z[nt]);
            }
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a1c[nt][r] = fmaxf(z[nt][r], 0.f);
                P[(q * 4 + r) * LDH + nt * 16 + j] = a1c[nt][r];
            }
        __syncthreads();
        // ---- recompute layer 2
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { const float b = sB2[nt * 16 + j]; z[nt] = (f32x4){b, b, b, b}; }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float x = P[j * LDH + 4 * kk + q];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) z[nt] = MFMA4(x, sW2[(nt * 16 + j) * LDH + 4 * kk + q], z[nt]);
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                a2c[nt][r] = fmaxf(z[nt][r], 0.f);
                Q[(q * 4 + r) * LDH + nt * 16 + j] = a2c[nt][r];
            }
        __syncthreads();
        // ---- dZ3 in A layout: lane (s = j, k = 4kk + q), kk = 0,1
        float dz3[2];
        {
            const int64_t m = m0 + j;
            const bool in = m < M;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int k = 4 * kk + q;
                float v = 0.f;
                if (in && k < a.out_dim) {
                    if (k == 0) v = dsigmas[m] * fminf(sigmas[m], e15);
                    else v = drgbs[m * nrgb + (k - 1)];
                }
                dz3[kk] = v;
                gb3[kk] += v;  // reduced over the 16 samples (lanes j) at the end
            }
        }
        // ---- dW3[n3][h] += sum_s dZ3[s][n3] A2[s][h] : A[i = n3][k = s] from global, B = A2 rows
        {
            const int k = j;  // output row n3
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int64_t m = m0 + 4 * kk + q;
                float v = 0.f;
                if (m < M && k < a.out_dim) {
                    if (k == 0) v = dsigmas[m] * fminf(sigmas[m], e15);
                    else v = drgbs[m * nrgb + (k - 1)];
                }
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) gW3[nt] = MFMA4(v, Q[(4 * kk + q) * LDH + nt * 16 + j], gW3[nt]);
            }
        }
        // ---- dA2 = dZ3 W3 ; dZ2 = dA2 * (A2 > 0)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) z[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) z[nt] = MFMA4(dz3[kk], sW3[(4 * kk + q) * LDH + nt * 16 + j], z[nt]);
        __syncthreads();  // all reads of A2 from Q are done
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = a2c[nt][r] > 0.f ? z[nt][r] : 0.f;
                Q[(q * 4 + r) * LDH + nt * 16 + j] = d;  // Q now holds dZ2 [s][o]
                gb2[nt] += d;
            }
        __syncthreads();
        // ---- dW2[o][i] += sum_s dZ2[s][o] A1[s][i]
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                av[t] = Q[(4 * kk + q) * LDH + t * 16 + j];
                bv[t] = P[(4 * kk + q) * LDH + t * 16 + j];
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) gW2[mt][nt] = MFMA4(av[mt], bv[nt], gW2[mt][nt]);
        }
        // ---- dA1 = dZ2 W2 : A from Q (sample on lanes), B[k = o][n = i] = W2[o][i]
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) z[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float x = Q[j * LDH + 4 * kk + q];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) z[nt] = MFMA4(x, sW2[(4 * kk + q) * LDH + nt * 16 + j], z[nt]);
        }
        __syncthreads();  // all reads of A1 from P are done
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = a1c[nt][r] > 0.f ? z[nt][r] : 0.f;
                P[(q * 4 + r) * LDH + nt * 16 + j] = d;  // P now holds dZ1 [s][o]
                gb1[nt] += d;
            }
        __syncthreads();
        // ---- dW1[o][i] += sum_s dZ1[s][o] X[s][i]
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            float av[4], bv[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) av[t] = P[(4 * kk + q) * LDH + t * 16 + j];
#pragma unroll
            for (int t = 0; t < 2; ++t) bv[t] = X[(4 * kk + q) * LDX + t * 16 + j];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                gW1[mt][0] = MFMA4(av[mt], bv[0], gW1[mt][0]);
                gW1[mt][1] = MFMA4(av[mt], bv[1], gW1[mt][1]);
            }
        }
        // ---- dX = dZ1 W1 : B[k = o][n = i] = W1[o][i]
        f32x4 dx[2];
        dx[0] = dx[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            const float x = P[j * LDH + 4 * kk + q];
            dx[0] = MFMA4(x, sW1[(4 * kk + q) * LDX + j], dx[0]);
            dx[1] = MFMA4(x, sW1[(4 * kk + q) * LDX + 16 + j], dx[1]);
        }
        // lane holds dX[s = q*4 + r][i = nt*16 + j] -> level-major dfeat
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int i = nt * 16 + j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = m0 + q * 4 + r;
                if (m < M) dfeat[((int64_t)(i >> 1) * a.level_stride + m) * 2 + (i & 1)] = dx[nt][r];
            }
        }
    }

    // ---- the four waves add their accumulators into one LDS slab in wave order (deterministic),
    //      then the workgroup writes ONE slab; k_mlp_reduce_slabs sums the slabs of all workgroups.
    float *slab = smem + O_X;
    __syncthreads();
    for (int i = tid; i < SLAB; i += 256) slab[i] = 0.f;
    __syncthreads();
    for (int turn = 0; turn < 4; ++turn) {
        if (w == turn) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int o = mt * 16 + q * 4 + r;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) slab[SL_W2 + o * HID + nt * 16 + j] += gW2[mt][nt][r];
                    slab[SL_W1 + o * IN + j] += gW1[mt][0][r];
                    slab[SL_W1 + o * IN + 16 + j] += gW1[mt][1][r];
                }
            // dW3: C rows = n3 (q*4 + r), cols = h (nt*16 + j)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) slab[SL_W3 + (q * 4 + r) * HID + nt * 16 + j] += gW3[nt][r];
            // biases: gb1/gb2 hold per-lane sums over this lane's 4 rows; add the 4 lane groups (q)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                float v1 = gb1[nt], v2 = gb2[nt];
                v1 += __shfl_xor(v1, 16, 64); v1 += __shfl_xor(v1, 32, 64);
                v2 += __shfl_xor(v2, 16, 64); v2 += __shfl_xor(v2, 32, 64);
                if (q == 0) { slab[SL_B1 + nt * 16 + j] += v1; slab[SL_B2 + nt * 16 + j] += v2; }
            }
            // gb3[kk]: lane (s = j, k = 4kk + q) -> sum over the 16 lanes j
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                float v = gb3[kk];
                v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
                v += __shfl_xor(v, 8, 64);
                if (j == 0) slab[SL_B3 + 4 * kk + q] += v;
            }
        }
        __syncthreads();
    }
    float *gslab = slabs + (int64_t)blockIdx.x * SLAB;
    for (int i = tid; i < SLAB; i += 256) gslab[i] = slab[i];
}

// 16 parameters per workgroup (421 workgroups: the chip is filled and a thread's chain of dependent adds is 32 long,
// not 128); 16 lane groups each sum every 16th slab, then the partials are added in a fixed order: deterministic.
constexpr int RED_P = 16, RED_G = 16;
__global__ void __launch_bounds__(256)
k_mlp_reduce_slabs(const float *__restrict__ slabs, int n_slabs, int out_dim, int accumulate,
                   float *__restrict__ dw1, float *__restrict__ db1, float *__restrict__ dw2, float *__restrict__ db2,
                   float *__restrict__ dw3, float *__restrict__ db3, uint32_t *__restrict__ clear, int clear_words) {
    // (a caller-named region zeroed on the side: the fill the next launch would otherwise need a dispatch for)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < clear_words; i += gridDim.x * 256) clear[i] = 0u;
    __shared__ float part[RED_G][RED_P];
    const int pi = threadIdx.x & (RED_P - 1), sg = threadIdx.x / RED_P;
    const int p = blockIdx.x * RED_P + pi;
    float s = 0.f;
    if (p < SLAB) {
#pragma unroll 8
        for (int b = sg; b < n_slabs; b += RED_G) s += slabs[(int64_t)b * SLAB + p];
    }
    part[sg][pi] = s;
    __syncthreads();
    if (sg != 0 || p >= SLAB) return;
    s = part[0][pi];
#pragma unroll
    for (int g = 1; g < RED_G; ++g) s += part[g][pi];
    float *dst = nullptr;
    if (p < SL_B1) dst = dw1 + (p - SL_W1);
    else if (p < SL_W2) dst = db1 + (p - SL_B1);
    else if (p < SL_B2) dst = dw2 + (p - SL_W2);
    else if (p < SL_W3) dst = db2 + (p - SL_B2);
    else if (p < SL_B3) { if ((p - SL_W3) / HID < out_dim) dst = dw3 + (p - SL_W3); }
    else { if (p - SL_B3 < out_dim) dst = db3 + (p - SL_B3); }
    if (dst) *dst = accumulate ? *dst + s : s;
}

int g_mlp_fwd_blocks = 768;   // 3 per CU: measured 24.5 us vs 28.1 (512) and 29.8 (1024) at the bench step
int g_mlp_fwd_wps = 3;   // wavefronts per SIMD the bf16 forward is compiled for: 3 (768 workgroups = 3 per CU; the
                         // register allocation is held at <= 168 so that they are co-resident), 2 or 4
int g_mlp_bwd_variant = 0;  // bf16 backward: 0 = shared staging images (barriers), 1 / 2 = operand-swap form at 1 / 2 waves per SIMD
int g_mlp_bwd_blocks = MLP_BWD_MAX_BLOCKS;  // persistent workgroups of the backward (<= MLP_BWD_MAX_BLOCKS slabs)

static int mlp_common_checks(const char *who, const void *feat, int feat_dtype, int64_t level_stride, const float *xyzs,
                             const float *w1, const float *b1, const float *w2, const float *b2, const float *w3,
                             const float *b3, int out_dim, float blob_std, int64_t m_host, int precision) {
    LNERF_REQUIRE(m_host >= 0 && level_stride >= m_host, "%s: need 0 <= m_host <= level_stride", who);
    // (the bf16 kernels address features and feature gradients with 32-bit byte offsets: 16 levels x level_stride x 8 B
    // must stay below 2^32; the exact-f32 kernels use 64-bit addressing and take any stride the scatter accepts)
    LNERF_REQUIRE(precision != LNERF_BF16 || level_stride <= ((int64_t)1 << 24),
                  "%s: level_stride must be <= 2^24 samples with the bf16 MLP (32-bit byte offsets)", who);
    LNERF_REQUIRE(level_stride < ((int64_t)1 << 30), "%s: level_stride must be below 2^30 samples", who);
    LNERF_REQUIRE(out_dim >= 2 && out_dim <= 8, "%s: out_dim must be in [2,8] (got %d)", who, out_dim);
    LNERF_REQUIRE(feat_dtype == LNERF_F32 || feat_dtype == LNERF_BF16, "%s: bad feat dtype", who);
    LNERF_REQUIRE(precision == LNERF_F32 || precision == LNERF_BF16, "%s: bad precision tag", who);
    LNERF_REQUIRE(blob_std > 0.f, "%s: blob_std must be > 0", who);
    if (m_host > 0) LNERF_REQUIRE(feat && xyzs && w1 && b1 && w2 && b2 && w3 && b3, "%s: null pointer", who);
    return LNERF_OK;
}

}  // namespace lnerf

using namespace lnerf;

extern "C" {

int lnerf_mlp_fragment_maps(int out_dim, int32_t *map_w1, int32_t *map_w2, int32_t *map_w3, lnerf_stream_t stream) {
    LNERF_REQUIRE(out_dim >= 2 && out_dim <= 8, "mlp_fragment_maps: out_dim must be in [2,8] (got %d)", out_dim);
    LNERF_REQUIRE(map_w1 && map_w2 && map_w3, "mlp_fragment_maps: null pointer");
    return launch_mlp_fragment_maps(out_dim, map_w1, map_w2, map_w3, as_stream(stream));
}

int lnerf_mlp_forward(const void *feat, int feat_dtype, int64_t level_stride, const float *xyzs, const float *w1,
                      const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, int out_dim,
                      float blob_scale, float blob_std, int64_t m_host, const int32_t *m_dev, float *sigmas,
                      float *rgbs, int precision, void *workspace, size_t workspace_bytes, lnerf_stream_t stream) {
    // (the caller keeps the fragments at the head of `workspace` current: the optimiser's fragment shadow, or an
    // earlier forward with the same weights)
    const bool fragments_ready = (precision & LNERF_MLP_FRAGMENTS_READY) != 0;
    precision &= ~LNERF_MLP_FRAGMENTS_READY;
    int rc = mlp_common_checks("mlp_forward", feat, feat_dtype, level_stride, xyzs, w1, b1, w2, b2, w3, b3, out_dim,
                               blob_std, m_host, precision);
    if (rc) return rc;
    if (m_host == 0) return LNERF_OK;
    LNERF_REQUIRE(sigmas && rgbs, "mlp_forward: null output");
    LNERF_REQUIRE(precision != LNERF_BF16 || out_dim != 5 || ((uintptr_t)rgbs & 15) == 0,
                  "mlp_forward: rgbs must be 16-byte aligned (one row per store)");
    MlpArgs a{feat, feat_dtype == LNERF_BF16, level_stride, xyzs, w1, b1, w2, b2, w3, b3, out_dim, blob_scale,
              2.0f * blob_std * blob_std, m_host, m_dev, nullptr};
    if (precision == LNERF_BF16) {
        // persistent workgroups walk ~M/128/blocks tiles each; with a workspace the weight fragments are built once per
        // launch (first MLP_FRAG_BYTES of it) instead of once per workgroup
        int64_t blocks = div_up(m_host, 128);
        if (blocks > g_mlp_fwd_blocks) blocks = g_mlp_fwd_blocks;
        if (workspace && workspace_bytes >= MLP_FRAG_BYTES) {
            LNERF_REQUIRE(((uintptr_t)workspace & 15) == 0, "mlp_forward: workspace must be 16-byte aligned");
            // all of them, the backward's too: lnerf_mlp_backward(... | LNERF_MLP_FRAGMENTS_READY) skips its own build
            const int rcf = fragments_ready ? LNERF_OK : launch_mlp_fragments_bf16(a, workspace, true, as_stream(stream));
            if (rcf) return rcf;
            a.frag_global = workspace;
        } else {
            LNERF_REQUIRE(!fragments_ready, "mlp_forward: LNERF_MLP_FRAGMENTS_READY without a workspace");
        }
        return launch_mlp_forward_bf16(a, sigmas, rgbs, (int)blocks, g_mlp_fwd_wps, as_stream(stream));
    }
    int64_t blocks = div_up(m_host, 64);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(k_mlp_forward_f32, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a, sigmas, rgbs);
    LNERF_CHECK_LAUNCH("mlp_forward");
    return LNERF_OK;
}

#ifdef LNERF_STAMPS
// diagnostic builds only: read (and clear) the per-phase shader-clock totals of the bf16 MLP kernels
int lnerf_debug_mlp_stamps(unsigned long long *out32) { return lnerf::mlp_stamps_read(out32); }
#endif

int lnerf_mlp_backward_slabs(int64_t m_host, int precision) {
    int64_t blocks = div_up(m_host, (precision & 0xFF) == LNERF_BF16 ? 128 : 64);
    if (blocks > BWD_MAX_BLOCKS) blocks = BWD_MAX_BLOCKS;
    if (blocks > g_mlp_bwd_blocks) blocks = g_mlp_bwd_blocks;
    return (int)(blocks < 0 ? 0 : blocks);
}

size_t lnerf_mlp_backward_workspace_bytes(int out_dim) {
    (void)out_dim;
    return MLP_FRAG_BYTES + (size_t)BWD_MAX_BLOCKS * SLAB * sizeof(float);  // fragment cache, then the slabs
}

int lnerf_mlp_backward(const void *feat, int feat_dtype, int64_t level_stride, const float *xyzs, const float *w1,
                       const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, int out_dim,
                       float blob_scale, float blob_std, int64_t m_host, const int32_t *m_dev, const float *sigmas,
                       const float *dsigmas, const float *drgbs, float *dfeat, float *dw1, float *db1, float *dw2,
                       float *db2, float *dw3, float *db3, int accumulate, void *workspace, size_t workspace_bytes,
                       int precision, void *clear_ptr, size_t clear_bytes, lnerf_stream_t stream) {
    const bool fragments_ready = (precision & LNERF_MLP_FRAGMENTS_READY) != 0;
    // LNERF_MLP_DEFER_REDUCE: the per-workgroup gradient slabs stay in the workspace -- lnerf_step_tail sums them and
    // applies the Adam step (the d* outputs are not written and may be NULL)
    const bool defer_reduce = (precision & LNERF_MLP_DEFER_REDUCE) != 0;
    precision &= ~(LNERF_MLP_FRAGMENTS_READY | LNERF_MLP_DEFER_REDUCE);
    int rc = mlp_common_checks("mlp_backward", feat, feat_dtype, level_stride, xyzs, w1, b1, w2, b2, w3, b3, out_dim,
                               blob_std, m_host, precision);
    if (rc) return rc;
    if (m_host == 0) return LNERF_OK;
    LNERF_REQUIRE(sigmas && dsigmas && drgbs && dfeat && (defer_reduce || (dw1 && db1 && dw2 && db2 && dw3 && db3)),
                  "mlp_backward: null pointer");
    LNERF_REQUIRE(!defer_reduce || clear_bytes == 0, "mlp_backward: the deferred form has no reduction launch to clear with");
    LNERF_REQUIRE(workspace && workspace_bytes >= lnerf_mlp_backward_workspace_bytes(out_dim),
                  "mlp_backward: workspace too small (%zu < %zu)", workspace_bytes,
                  lnerf_mlp_backward_workspace_bytes(out_dim));
    LNERF_REQUIRE(((uintptr_t)workspace & 15) == 0, "mlp_backward: workspace must be 16-byte aligned");
    LNERF_REQUIRE((clear_bytes == 0 || clear_ptr) && (clear_bytes & 3) == 0 && ((uintptr_t)clear_ptr & 3) == 0 &&
                  clear_bytes <= ((size_t)1 << 24), "mlp_backward: bad clear region");
    LNERF_REQUIRE(precision != LNERF_BF16 || out_dim != 5 || ((uintptr_t)drgbs & 15) == 0,
                  "mlp_backward: drgbs must be 16-byte aligned (one row per load)");
    MlpArgs a{feat, feat_dtype == LNERF_BF16, level_stride, xyzs, w1, b1, w2, b2, w3, b3, out_dim, blob_scale,
              2.0f * blob_std * blob_std, m_host, m_dev, nullptr};
    const int64_t blocks = lnerf_mlp_backward_slabs(m_host, precision);
    hipStream_t s = as_stream(stream);
    float *slabs = reinterpret_cast<float *>(static_cast<char *>(workspace) + MLP_FRAG_BYTES);
    if (precision == LNERF_BF16) {
        int rc2 = fragments_ready ? LNERF_OK : launch_mlp_fragments_bf16(a, workspace, true, s);   // once per launch, not once per workgroup
        if (rc2) return rc2;
        a.frag_global = workspace;
        rc2 = launch_mlp_backward_bf16(a, sigmas, dsigmas, drgbs, dfeat, slabs, (int)blocks, g_mlp_bwd_variant, s);
        if (rc2) return rc2;
    } else {
        hipLaunchKernelGGL(k_mlp_backward_f32, dim3((unsigned)blocks), dim3(256), 0, s, a, sigmas, dsigmas, drgbs,
                           dfeat, slabs);
        LNERF_CHECK_LAUNCH("mlp_backward");
    }
    if (defer_reduce) return LNERF_OK;
    hipLaunchKernelGGL(k_mlp_reduce_slabs, dim3((unsigned)div_up(SLAB, RED_P)), dim3(256), 0, s,
                       (const float *)slabs, (int)blocks, out_dim, accumulate, dw1, db1, dw2, db2, dw3, db3,
                       (uint32_t *)clear_ptr, (int)(clear_bytes / 4));
    LNERF_CHECK_LAUNCH("mlp_backward(reduce)");
    return LNERF_OK;
}

}  // extern "C"
