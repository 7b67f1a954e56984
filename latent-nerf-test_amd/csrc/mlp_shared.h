// Shared between the f32 (mlp.hip) and bf16 (mlp_bf16.hip) MLP kernels.
#pragma once
#include "common.h"

namespace lnerf {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int MLP_IN = 32, MLP_HID = 64, MLP_OUTP = 16;  // OUTP: padded output width (one 16-wide tile)

// gradient slab layout (floats): dW1 [64*32] | db1 [64] | dW2 [64*64] | db2 [64] | dW3 [16*64] | db3 [16]
constexpr int MLP_SL_W1 = 0, MLP_SL_B1 = MLP_SL_W1 + MLP_HID * MLP_IN, MLP_SL_W2 = MLP_SL_B1 + MLP_HID,
              MLP_SL_B2 = MLP_SL_W2 + MLP_HID * MLP_HID, MLP_SL_W3 = MLP_SL_B2 + MLP_HID,
              MLP_SL_B3 = MLP_SL_W3 + MLP_OUTP * MLP_HID, MLP_SLAB = MLP_SL_B3 + MLP_OUTP;
constexpr int MLP_BWD_MAX_BLOCKS = 512;

struct MlpArgs {
    const void *feat;
    int feat_bf16;
    int64_t level_stride;
    const float *xyzs;
    const float *w1, *b1, *w2, *b2, *w3, *b3;
    int out_dim;
    float blob_scale, blob_denom;  // blob = scale * exp(-|x|^2 / denom), denom = 2 std^2
    int64_t m_host;
    const int32_t *m_dev;
    const void *frag_global;  // bf16 path: the weight fragments, built once per launch (NULL: every workgroup builds its own)
};
constexpr size_t MLP_FRAG_BYTES = LNERF_MLP_FRAGMENT_BYTES;  // room for the 33 one-KiB fragments of the bf16 path at the head of a workspace

__device__ __forceinline__ float blob_of(const MlpArgs &a, int64_t m) {
    const float x = a.xyzs[m * 3], y = a.xyzs[m * 3 + 1], z = a.xyzs[m * 3 + 2];
    const float d2 = (x * x + y * y) + z * z;
    return a.blob_scale * expf(-d2 / a.blob_denom);
}

// bf16 path launchers (mlp_bf16.hip)
int launch_mlp_fragments_bf16(const MlpArgs &a, void *frag_out, bool backward_too, hipStream_t stream);
int launch_mlp_fragment_maps(int out_dim, int32_t *m1, int32_t *m2, int32_t *m3, hipStream_t stream);
int launch_mlp_forward_bf16(const MlpArgs &a, float *sigmas, float *rgbs, int blocks, int wps, hipStream_t stream);
int launch_mlp_backward_bf16(const MlpArgs &a, const float *sigmas, const float *dsigmas, const float *drgbs,
                             float *dfeat, float *slabs, int blocks, int variant, hipStream_t stream);

#ifdef LNERF_STAMPS
int mlp_stamps_read(unsigned long long *out32);   // diagnostic builds only (mlp_bf16.hip)
#endif

}  // namespace lnerf
