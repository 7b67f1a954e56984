/* lnerf_hip.h -- C ABI of the MI355X (gfx950) latent-NeRF render kernels.
 *
 * This is the drop-in boundary of the hot path named by BASELINE.json `north_star`
 * (SURVEY.md §8): everything `NeRFRenderer.render()/run_cuda()` needs per optimisation step.
 * The reference checkout calls this path at scripts/train_latent_nerf.py:3-4,10-14
 * (`from src.latent_nerf... import Trainer` -> trainer -> renderer) but does not contain it
 * (README.md:152-156 lists `src/latent_nerf/raymarching`, "The CUDA ray marching modules");
 * each entry point below names the extension op of that absent module it stands in for, as
 * enumerated in SURVEY.md §8(b).  The caller-side contract that IS present in the reference --
 * the renderer hands the trainer `{'image': [B,4,H,W]}` and receives the SDS gradient through
 * `pred.backward(gradient=grad)` -- is src/latent_paint/models/textured_mesh.py:181-220 and
 * src/latent_paint_mesh/training/trainer.py:657-658; the Python host side above this ABI
 * (latent-nerf-test_amd/src/latent_nerf) honours it.
 *
 * Conventions
 *   - plain C: device pointers, sizes, a stream handle.  No torch types.  All pointers are
 *     DEVICE pointers unless the name ends in `_host`.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Every call only
 *     enqueues work; none synchronises, allocates or frees (all are hipGraph-capturable).
 *   - return value: LNERF_OK (0) or a negative LNERF_ERR_*; lnerf_last_error() gives a
 *     thread-local message.  Arguments are validated on the host before any launch.
 *   - data-dependent sizes (the number of samples M a march produced) stay on the device:
 *     kernels that consume samples take `m_host` (an upper bound, usually the buffer
 *     capacity) and an optional device pointer `m_dev`; they process min(m_host, *m_dev).
 *   - sample-major feature tensors are stored LEVEL-MAJOR: feat[(l*F+f)] lives at
 *     base + (l * level_stride + m) * F + f  (F = 2), so every wave writes/reads 512 contiguous
 *     bytes per level.  `level_stride` is in samples (normally the buffer capacity).
 */
#ifndef LNERF_HIP_H
#define LNERF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LNERF_ABI_VERSION 7

#define LNERF_OK 0
#define LNERF_ERR_INVALID_ARG (-1)
#define LNERF_ERR_HIP (-2)
#define LNERF_ERR_UNSUPPORTED (-3)

/* dtype tags for `void*` tensors */
#define LNERF_F32 0
#define LNERF_BF16 1

#define LNERF_MAX_LEVELS 32

typedef void *lnerf_stream_t;

int lnerf_abi_version(void);
const char *lnerf_last_error(void);
/* "gfx950;<git or build tag>" -- lets the host side assert what it loaded */
const char *lnerf_build_info(void);

/* Performance knobs (never change results beyond float summation order).  Keys:
 *   "scatter_compact_max_res": levels with resolution <= value merge per-wavefront runs of samples in one cell
 *                              before binning (default 512).
 *   "scatter_bin_per_cu":      persistent workgroups of the binning pass per CU, 1 .. 4 (default 3).
 *   "scatter_bin_wgs":         persistent workgroups of the binning pass (default 0 = 256 x scatter_bin_per_cu).
 *   "scatter_skip_zero":       1 (default) = contributions that are exactly zero are not binned.
 *   "scatter_reduce_threads":  threads per workgroup of the reduce pass, 512 or 1024 (default 1024).
 *   "gather_pair_loads":       0 = one load per vertex; 1 = x-adjacent vertices of dense levels with one load;
 *                              2 (default) = additionally one aligned 16-byte load per 4-row group of a bf16 table.
 *   "gather_dedup_max_res":    levels with resolution <= value fetch a cell's 8 vertices once per run of
 *                              lanes (consecutive samples of a ray) in that cell (default 512; 0 = off).
 *   "mlp_fwd_blocks":          persistent workgroups of the bf16 MLP forward (default 768).
 *   "mlp_fwd_wps":             wavefronts per SIMD the bf16 forward is compiled for, 3 (default), 2 or 4.
 *   "gather_wgs_per_xcd":      workgroups per XCD of the XCD-owned-level gather variant (variant 2).
 *   "mlp_bwd_blocks":          persistent workgroups of the MLP backward (default and maximum 512 = slab count).
 */
int lnerf_set_tuning(const char *key, int value);

/* ---- H1: ray generation (absent upstream: `get_rays` of nerf_utils; camera convention
 * src/latent_paint/models/render.py:19-31).  c2w [B,4,4] row-major, columns (right,down,forward,eye).
 * rays_o, rays_d [B, H*W, 3]. */
int lnerf_get_rays(const float *c2w, int B, int H, int W, float fx, float fy, float cx, float cy,
                   float *rays_o, float *rays_d, lnerf_stream_t stream);

/* ---- H2: `raymarching.near_far_from_aabb`.  aabb = (xmin,ymin,zmin,xmax,ymax,zmax) by value.
 * Misses get near = far = FLT_MAX. */
int lnerf_near_far_from_aabb(const float *rays_o, const float *rays_d, int64_t N, float xmin, float ymin, float zmin,
                             float xmax, float ymax, float zmax, float min_near, float *nears, float *fars,
                             lnerf_stream_t stream);

/* ---- H3: `raymarching.morton3D`, `morton3D_invert`, `packbits`. */
int lnerf_morton3d(const int32_t *coords, int64_t n, uint32_t *indices, lnerf_stream_t stream);
int lnerf_morton3d_invert(const uint32_t *indices, int64_t n, int32_t *coords, lnerf_stream_t stream);
/* bit k of byte b = grid[8b+k] > min(thresh, *mean_dev)  (mean_dev may be NULL) */
int lnerf_packbits(const float *grid, int64_t n_cells, float thresh, const float *mean_dev, uint8_t *bitfield,
                   lnerf_stream_t stream);

/* ---- H4: `raymarching.march_rays_train`.
 * Two launches up to 8192 rays: per-ray count (one wavefront per ray, ballot + popcount over 64 lattice points at a
 * time), per-ray write (ballot prefix-sum compaction) whose wavefront first sums the counts of the rays before its own;
 * three launches above (count, single-workgroup exclusive scan, write).  Output order is deterministic: samples of ray
 * n follow those of ray n-1.
 *   rays    int32 [N,3]  (ray id, offset, count)
 *   counter int32 [lnerf_march_counter_len(N)]
 *                        [0]=M total samples written, [1]=number of rays with count>0,
 *                        [2]=rays dropped because offset+count exceeded `capacity`,
 *                        [3]=running maximum of  M | (rays dropped ? 2^30 : 0)  over the calls since the CALLER last
 *                        zeroed it (zero it when the buffer is allocated): the peak a training loop sizes its sample
 *                        buffers from, kept without a launch of its own,
 *                        [4 ...) scratch of the call (two-launch form: per-ray counts, per-workgroup sums)
 *   jitter of the march start t0 = near + dt(near) * u_n, one of
 *     noises [N] in [0,1)          the upstream form (`noises = torch.rand(N)`);
 *     noise_counter (device int32) counter-based generator: u_n = hash(n, noise_seed, *noise_counter) in [0,1)
 *                                  (24 bits; the hash is restated in oracle/nerf_oracle.py `march_noise`), and the
 *                                  call advances *noise_counter by one -- fresh jitter on every replay of a
 *                                  captured hipGraph, no host RNG state;
 *     both NULL                    no jitter.
 *   xyzs [capacity,3], dirs [capacity,3], deltas [capacity,2] = (dt, t). */
int64_t lnerf_march_counter_len(int64_t N);
int lnerf_march_rays_train(const float *rays_o, const float *rays_d, const float *nears, const float *fars, int64_t N,
                           const uint8_t *bitfield, float bound, int cascade, int grid_size, int max_steps,
                           float dt_gamma, const float *noises, uint32_t noise_seed, int32_t *noise_counter,
                           int64_t capacity, float *xyzs, float *dirs, float *deltas, int32_t *rays, int32_t *counter,
                           lnerf_stream_t stream);
/* The same with the AABB clip of lnerf_near_far_from_aabb done inside the two march passes (identical arithmetic):
 * the training path of `NeRFRenderer.run_cuda` (SURVEY.md §8(a) H2 -> H4, line 465) calls near_far_from_aabb only
 * to feed march_rays_train, and in a replayed graph every dispatch of a few thousand rays costs ~4.5 us whatever it
 * computes. */
int lnerf_march_rays_train_aabb(const float *rays_o, const float *rays_d, float xmin, float ymin, float zmin, float xmax,
                                float ymax, float zmax, float min_near, int64_t N, const uint8_t *bitfield, float bound,
                                int cascade, int grid_size, int max_steps, float dt_gamma, const float *noises,
                                uint32_t noise_seed, int32_t *noise_counter, int64_t capacity, float *xyzs, float *dirs,
                                float *deltas, int32_t *rays, int32_t *counter, lnerf_stream_t stream);
/* ... and with the ray generation of lnerf_get_rays folded into the count pass as well (same arithmetic): the rays of B
 * views of H x W pixels are generated, written to rays_o_out / rays_d_out [B*H*W, 3] (the caller's background and
 * direction-dependent heads read them there) and marched.  One more dispatch off the step. */
int lnerf_march_rays_train_pose(const float *c2w, int B, int H, int W, float fx, float fy, float cx, float cy,
                                float *rays_o_out, float *rays_d_out, float xmin, float ymin, float zmin, float xmax,
                                float ymax, float zmax, float min_near, const uint8_t *bitfield, float bound, int cascade,
                                int grid_size, int max_steps, float dt_gamma, const float *noises, uint32_t noise_seed,
                                int32_t *noise_counter, int64_t capacity, float *xyzs, float *dirs, float *deltas,
                                int32_t *rays, int32_t *counter, lnerf_stream_t stream);
/* The same with the intrinsics in DEVICE memory: intrinsics f32 [B,4] = (fx, fy, cx, cy) per view.  Nothing about the
 * camera is baked into the launch, so a captured hipGraph renders whatever pose / field of view its caller copied into
 * `c2w` / `intrinsics` before the replay (the trainer's graphed step: a new random view every step). */
int lnerf_march_rays_train_camera(const float *c2w, const float *intrinsics, int B, int H, int W, float *rays_o_out,
                                  float *rays_d_out, float xmin, float ymin, float zmin, float xmax, float ymax,
                                  float zmax, float min_near, const uint8_t *bitfield, float bound, int cascade,
                                  int grid_size, int max_steps, float dt_gamma, const float *noises, uint32_t noise_seed,
                                  int32_t *noise_counter, int64_t capacity, float *xyzs, float *dirs, float *deltas,
                                  int32_t *rays, int32_t *counter, lnerf_stream_t stream);

/* ---- H4 (inference): `raymarching.march_rays` / `composite_rays` and the live-ray compaction
 * the upstream renderer does on the host (`rays_alive = rays_alive[rays_alive >= 0]`). */
int lnerf_march_rays(int64_t n_alive, int n_step, const int32_t *rays_alive, const float *rays_t, const float *rays_o,
                     const float *rays_d, const float *fars, const uint8_t *bitfield, float bound, int cascade,
                     int grid_size, int max_steps, float dt_gamma, float *xyzs, float *dirs, float *deltas,
                     lnerf_stream_t stream);
int lnerf_composite_rays(int64_t n_alive, int n_step, int32_t *rays_alive, float *rays_t, const float *sigmas,
                         const float *rgbs, const float *deltas, int C, float T_thresh, float *weights_sum,
                         float *depth, float *image, float *transmittance, lnerf_stream_t stream);
/* alive_out[0..n) = entries of alive_in that are >= 0, order preserved; n -> *n_alive_dev.
 * Wave ballot + prefix-sum compaction, one launch. */
int lnerf_compact_rays(const int32_t *alive_in, int64_t n, int32_t *alive_out, int32_t *n_alive_dev,
                       lnerf_stream_t stream);

/* ---- H5/H6: `gridencoder.grid_encode_forward/backward` (multiresolution hash grid, F = 2).
 * Level metadata is passed from the host (num_levels <= LNERF_MAX_LEVELS):
 *   offsets_host [L+1] row offsets, scales_host [L] per-level scale, res_host [L] resolution.
 * xyzs are world positions; the kernels normalise x01 = (x + bound) / (2*bound).
 * `variant` selects the workgroup->(level,tile) mapping: 0 = level on blockIdx.y,
 * 1 = XCD-aware (levels pinned to XCDs so each XCD's L2 holds two levels). */
int lnerf_grid_encode_forward(const float *xyzs, float bound, const void *table, int table_dtype, int num_levels,
                              int level_dim, const int32_t *offsets_host, const float *scales_host,
                              const int32_t *res_host, int64_t m_host, const int32_t *m_dev, int64_t level_stride,
                              void *feat, int feat_dtype, int variant, lnerf_stream_t stream);
/* dtable (f32, [rows, F]) is ACCUMULATED into (+=).
 * variant 0/1: per-lane global float atomics (blockIdx.y level map / XCD-aware map); no workspace.
 * variant 2  : two-pass bucketed scatter -- records binned per 64 KiB table chunk with plain
 *              stores, then reduced in LDS in 64-bit fixed point and added with coalesced stores
 *              (heavily loaded coarse chunks are cut into slices whose exact integer partial sums the
 *              slice that finishes last adds up): the sums do not depend on execution order, the
 *              gradient is bitwise reproducible.  Needs `workspace` of
 *              lnerf_grid_encode_backward_workspace_bytes() bytes (16-byte aligned) whose first
 *              LNERF_SCATTER_ZERO_HEAD_BYTES were zero when it was FIRST used (arrival counters that
 *              every call leaves zero again; the rest may hold anything).
 * variant 3  : variant 2 with packed 8-byte records (12-bit row inside the bucket + two values rounded
 *              to 26-bit floats, 17 mantissa bits): a third less record traffic, relative rounding
 *              2^-18 per addend; same workspace. */
#define LNERF_SCATTER_ZERO_HEAD_BYTES (64 * 1024)
size_t lnerf_grid_encode_backward_workspace_bytes(int num_levels, const int32_t *offsets_host, int64_t m_host);
/* The first lnerf_grid_scatter_clear_bytes() bytes of that workspace (bucket cursors, level maxima) are cleared by every
 * bucketed scatter call with a fill dispatch of its own -- ~5 us in a replayed graph for a few KiB.  A caller that
 * launches something right before the scatter anyway can clear them there (lnerf_mlp_backward takes such a region)
 * and pass `variant | LNERF_SCATTER_CLEARED`. */
#define LNERF_SCATTER_CLEARED 0x100
/* variant | LNERF_GRID_BLOCKED (forward AND backward entry points, consistently): an opt-in layout of the HASHED levels
 * (not Instant-NGP's): the vertex lattice is cut into blocks of 4 x 2 x 2, the block coordinate is hashed and the 16 rows
 * of a block are consecutive: row = (hash(x >> 2, y >> 1, z >> 1) mod (rows / 16)) * 16 + (x & 3) + 4 (y & 1) + 8 (z & 1).
 * One block = one 64-byte line of the bf16 table: a sample's 8 vertices touch 2.8 lines on average instead of 4.25.
 * Dense levels are unchanged.  Restated in oracle/nerf_oracle.py (grid_corner_indices(blocked=True)). */
#define LNERF_GRID_BLOCKED 0x400
/* variant | LNERF_GRID_TILED (forward AND backward entry points, consistently; not together with LNERF_GRID_BLOCKED): the
 * upstream encoder's `gridtype = "tiled"` -- a level too large for its table wraps its dense index,
 * row = (x + y (res + 1) + z (res + 1)^2 mod 2^32) mod rows, instead of hashing the vertex.  Restated in
 * oracle/nerf_oracle.py (grid_corner_indices(layout="tiled")). */
#define LNERF_GRID_TILED 0x800
/* variant | LNERF_SCATTER_DEFER_FINISH: accepted and ignored (ABI 4 deferred a separate finishing pass of the sliced
 * buckets to lnerf_step_tail; pass 2 finishes them itself now). */
#define LNERF_SCATTER_DEFER_FINISH 0x200
size_t lnerf_grid_scatter_clear_bytes(int num_levels, const int32_t *offsets_host, int64_t m_host);
int lnerf_grid_encode_backward(const float *xyzs, float bound, const void *dfeat, int dfeat_dtype, int num_levels,
                               int level_dim, const int32_t *offsets_host, const float *scales_host,
                               const int32_t *res_host, int64_t m_host, const int32_t *m_dev, int64_t level_stride,
                               float *dtable, int variant, void *workspace, size_t workspace_bytes,
                               lnerf_stream_t stream);
/* Backward of the hash grid with the gradient WRITTEN (not accumulated) as bf16 pairs, the wire format of
 * the data-parallel all-reduce: `grad_bf16` ([rows, 2] bf16) needs no zero fill and every row is written
 * exactly once by the kernel that finishes its sum -- no read-modify-write of an f32 table gradient and no
 * cast afterwards.  variant 2 or 3.  `dtable_zero` (f32 [rows, 2]) only carries the records of overflowing
 * buckets; it must be ZERO on entry and is zero again on return. */
int lnerf_grid_encode_backward_bf16(const float *xyzs, float bound, const void *dfeat, int dfeat_dtype, int num_levels,
                                    int level_dim, const int32_t *offsets_host, const float *scales_host,
                                    const int32_t *res_host, int64_t m_host, const int32_t *m_dev,
                                    int64_t level_stride, float *dtable_zero, int variant, void *workspace,
                                    size_t workspace_bytes, void *grad_bf16, lnerf_stream_t stream);
/* Split form of lnerf_grid_encode_backward_bf16 for a PIPELINED data-parallel exchange: pass 1 once for all levels
 * (lnerf_grid_scatter_bin; clears the bucket cursors), then pass 2 + the finishing pass per level range
 * (lnerf_grid_scatter_reduce_bf16: writes rows offsets[level_lo] .. offsets[level_hi] of grad_bf16), so that the
 * all-reduce of a level group can be launched while the next group is still being summed.  Same workspace, same
 * arithmetic, same bits as the one-call form. */
int lnerf_grid_scatter_bin(const float *xyzs, float bound, const void *dfeat, int dfeat_dtype, int num_levels,
                           int level_dim, const int32_t *offsets_host, const float *scales_host, const int32_t *res_host,
                           int64_t m_host, const int32_t *m_dev, int64_t level_stride, float *dtable_zero, int variant,
                           void *workspace, size_t workspace_bytes, lnerf_stream_t stream);
int lnerf_grid_scatter_reduce_bf16(float bound, int num_levels, int level_dim, const int32_t *offsets_host,
                                   const float *scales_host, const int32_t *res_host, int64_t m_host,
                                   int64_t level_stride, int level_lo, int level_hi, float *dtable_zero, int variant,
                                   void *workspace, size_t workspace_bytes, void *grad_bf16, lnerf_stream_t stream);

/* Backward of the hash grid fused with the table's optimiser step (single-GPU training: no gradient
 * exchange sits between the two).  Same scatter as above (variant 2 or 3), but the kernel that
 * finishes a row's sum (pass 2, or the finishing kernel of the sliced coarse levels) applies
 * Adam(beta1, beta2, eps) to it straight from the fixed-point sum -- `table`, `exp_avg`, `exp_avg_sq`
 * (f32 [rows, 2]) and the optional bf16 `shadow_bf16` are updated in place and the gradient never
 * reaches HBM.  `dtable_zero` (f32 [rows, 2]) only carries the records of overflowing buckets; it
 * must be ZERO on entry and is zero again on return.  Arithmetic and results are bit-identical to
 * lnerf_grid_encode_backward + lnerf_adam_step (one shared definition).
 * Covers `optimizer.zero_grad() ... optimizer.step()` of src/latent_paint/training/trainer.py:127-131
 * for the table parameter only. */
int lnerf_grid_encode_backward_adam(const float *xyzs, float bound, const void *dfeat, int dfeat_dtype, int num_levels,
                                    int level_dim, const int32_t *offsets_host, const float *scales_host,
                                    const int32_t *res_host, int64_t m_host, const int32_t *m_dev,
                                    int64_t level_stride, float *dtable_zero, int variant, void *workspace,
                                    size_t workspace_bytes, float *table, float *exp_avg, float *exp_avg_sq,
                                    void *shadow_bf16, float lr, float beta1, float beta2, float eps, int step,
                                    const int32_t *step_dev, float grad_scale, lnerf_stream_t stream);

/* ---- H7 helper: inverse of the bf16 weight-fragment layout at the head of the MLP workspace: map_wK[2 i], map_wK[2 i + 1] = the two
 * bf16 elements of that image which hold weight i of wK (forward / transposed fragments).  map_w1 int32[64*32*2],
 * map_w2 int32[64*64*2], map_w3 int32[out_dim*64*2].  For lnerf_adam_step_multi_shadow. */
int lnerf_mlp_fragment_maps(int out_dim, int32_t *map_w1, int32_t *map_w2, int32_t *map_w3, lnerf_stream_t stream);

/* ---- H7: fused sigma/latent MLP  32 -> 64 -> 64 -> out_dim (= 1 + C), ReLU hidden.
 * Weights are PyTorch nn.Linear layout: w1 [64,32], b1 [64], w2 [64,64], b2 [64], w3 [out_dim,64],
 * b3 [out_dim], all f32.  sigma = exp(h0 + blob_scale*exp(-|x|^2/(2 blob_std^2))), rgbs = h[1:].
 * precision: LNERF_F32 -> exact-f32 MFMA (v_mfma_f32_16x16x4_f32), LNERF_BF16 -> bf16 MFMA, f32 acc.
 * level_stride <= 2^24 samples with LNERF_BF16 (32-bit byte offsets inside the bf16 kernels; the exact-f32 kernels take
 * any stride below 2^30); with out_dim == 5 the bf16 path moves the
 * latent rows (rgbs, drgbs: [*, 4] f32) 16 bytes at a time: those buffers must be 16-byte aligned.
 * workspace (optional, 16-byte aligned, >= 36 KiB; the buffer of lnerf_mlp_backward_workspace_bytes() serves): with
 * it the bf16 path builds its weight fragments (the backward's too) once per launch instead of once per workgroup;
 * precision | LNERF_MLP_FRAGMENTS_READY: they are current already (fragment shadow of the optimiser, or an earlier
 * forward with the same weights) -- no build at all. */
int lnerf_mlp_forward(const void *feat, int feat_dtype, int64_t level_stride, const float *xyzs, const float *w1,
                      const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, int out_dim,
                      float blob_scale, float blob_std, int64_t m_host, const int32_t *m_dev, float *sigmas,
                      float *rgbs, int precision, void *workspace, size_t workspace_bytes, lnerf_stream_t stream);
/* Recomputes the hidden activations.  dfeat is written (level-major, f32); the d* parameter gradients
 * are accumulated (accumulate != 0: +=) or overwritten (accumulate == 0) deterministically:
 * per-workgroup partial slabs in `workspace` (lnerf_mlp_backward_workspace_bytes()) followed by one
 * reduction launch that sums them in a fixed order.
 * precision: LNERF_F32 or LNERF_BF16; LNERF_BF16 | LNERF_MLP_FRAGMENTS_READY says that the head of `workspace`
 * still holds the fragments lnerf_mlp_forward built from these very weights (same workspace, no weight update in
 * between), so the backward does not rebuild them.
 * clear_ptr / clear_bytes (may be NULL / 0; 4-byte granular): a small region the slab-reduction launch also zeroes
 * -- e.g. the cursors of the scatter that follows (LNERF_SCATTER_CLEARED): one dispatch less per step. */
#define LNERF_MLP_FRAGMENTS_READY 0x100
/* precision | LNERF_MLP_DEFER_REDUCE: the per-workgroup gradient slabs stay in `workspace` (lnerf_mlp_backward_slabs() of
 * them); the d* outputs are not written (may be NULL), clear_bytes must be 0.  lnerf_step_tail sums the slabs and applies
 * the Adam step of the six tensors. */
#define LNERF_MLP_DEFER_REDUCE 0x200
int lnerf_mlp_backward_slabs(int64_t m_host, int precision);
#define LNERF_MLP_FRAGMENT_BYTES (36 * 1024) /* the bf16 weight-fragment image at the head of the workspace */
size_t lnerf_mlp_backward_workspace_bytes(int out_dim);
int lnerf_mlp_backward(const void *feat, int feat_dtype, int64_t level_stride, const float *xyzs, const float *w1,
                       const float *b1, const float *w2, const float *b2, const float *w3, const float *b3, int out_dim,
                       float blob_scale, float blob_std, int64_t m_host, const int32_t *m_dev, const float *sigmas,
                       const float *dsigmas, const float *drgbs, float *dfeat, float *dw1, float *db1, float *dw2,
                       float *db2, float *dw3, float *db3, int accumulate, void *workspace, size_t workspace_bytes,
                       int precision, void *clear_ptr, size_t clear_bytes, lnerf_stream_t stream);

/* ---- trainer helper: gradient of the opacity-entropy regulariser of the NeRF trainer (sparsity term) w.r.t.
 * weights_sum [N], in one launch:  L = scale * mean_i H(clamp(ws_i, eps, 1 - eps)),  H(p) = -p log2 p - (1-p) log2(1-p);
 * grad_i = scale / N * (log2(1 - ws_i) - log2 ws_i) for eps <= ws_i <= 1 - eps, else 0.  The result is handed to the
 * compositing backward as grad_weights_sum. */
int lnerf_opacity_entropy_grad(const float *weights_sum, int64_t N, float scale, float eps, float *grad,
                               lnerf_stream_t stream);

/* ---- trainer helper: the SEEDED SYNTHETIC guidance (the stand-in for `grad = diffusion.train_step(text_z, pred)` of the
 * reference's src/stable_diffusion.py:248-334 where no diffusion model is available) and, optionally, the entropy gradient
 * above, in ONE launch and in the renderer's own image layout:
 *   t = t_lo + floor(u (t_hi - t_lo + 1)),  w = weights[t]   (weights f32 [>= t_hi + 1]: sqrt(a_t)(1 - a_t), :274, :320)
 *   grad_image[v, p, c] = w * (noise_scale * z + (image[v, p, c] - targets[dirs[v], p, c]))
 * image / grad_image f32 [n_views, rays_per_view, C]; targets f32 [n_buckets, rays_per_view, C]; dirs int32 [n_views]
 * (clamped into range).  u and the normal deviates z come from a counter-based generator of (seed, *step_dev, element)
 * -- *step_dev is only read (the optimiser's device step counter: it advances once per step), so a replayed hipGraph
 * draws fresh noise without host RNG state.  grad_weights_sum != NULL: also lnerf_opacity_entropy_grad over
 * weights_sum [n_views * rays_per_view].  Restated in oracle/nerf_oracle.py synthetic_guidance(). */
int lnerf_synthetic_guidance(const float *image, const float *targets, const int32_t *dirs, const float *weights,
                             int64_t n_views, int rays_per_view, int C, int n_buckets, int t_lo, int t_hi,
                             float noise_scale, uint32_t seed, const int32_t *step_dev, float *grad_image,
                             const float *weights_sum, float ent_scale, float ent_eps, float *grad_weights_sum,
                             lnerf_stream_t stream);

/* ---- H8/H9: `raymarching.composite_rays_train_forward/backward`.  One wavefront per ray,
 * log-space prefix scan of sigma*dt across lanes.  C = colour channels (3 or 4).
 * bg_color [N,C] or NULL: image += (1 - weights_sum) * bg. */
int lnerf_composite_rays_train_forward(const float *sigmas, const float *rgbs, const float *deltas,
                                       const int32_t *rays, int64_t N, int C, float T_thresh, const float *bg_color,
                                       float *weights_sum, float *depth, float *image, lnerf_stream_t stream);
/* grad_weights_sum / grad_depth / grad_bg may be NULL.  grad_sigmas [.], grad_rgbs [.,C] are
 * written for every sample inside a ray's span (zeros after early termination). */
int lnerf_composite_rays_train_backward(const float *grad_weights_sum, const float *grad_depth,
                                        const float *grad_image, const float *sigmas, const float *rgbs,
                                        const float *deltas, const int32_t *rays, const float *weights_sum,
                                        const float *depth, const float *image, const float *bg_color, int64_t N,
                                        int C, float T_thresh, float *grad_sigmas, float *grad_rgbs, float *grad_bg,
                                        lnerf_stream_t stream);

/* ---- H10: occupancy grid refresh pieces (`update_extra_state`): cell sample points,
 * decayed max update, mean, then lnerf_packbits.  Update and mean are ORDER-INDEPENDENT (replicas of a data-parallel
 * run refresh their grids redundantly and must stay bit-identical): a cell listed several times takes the maximum of
 * its new densities, the mean is summed in a fixed order. */
int lnerf_occ_cell_points(const uint32_t *indices, int64_t n, int cascade_level, int grid_size, float bound,
                          const float *noise, float *xyzs, lnerf_stream_t stream);
/* Steady-state cell sampling of the refresh, on the device: indices [2*n_rand] = n_rand random cells followed by n_rand
 * cells of the occupied ones (grid > 0; of all cells when none is), xyzs [2*n_rand, 3] = a jittered point in each (the
 * formula of lnerf_occ_cell_points).  The draws are STRATIFIED (ABI 7): draw j of a half takes one element, uniformly, from
 * the j-th of n_rand equal strata of its population (the cells in Morton order; the ascending list of occupied cells) --
 * the marginal probabilities of independent draws, ascending output, neighbouring cells on neighbouring lanes of the
 * density query that follows.  Random numbers: u = hash(i, seed, step, k), restated in oracle/nerf_oracle.py `occ_sample`.  The occupied list is built in ascending cell order and its length never visits
 * the host (the upstream form synchronises on torch.nonzero).  scratch: lnerf_occ_sample_scratch_bytes(n_cells). */
size_t lnerf_occ_sample_scratch_bytes(int64_t n_cells);
int lnerf_occ_sample(const float *grid_level, int64_t n_cells, int cascade_level, int grid_size, float bound,
                     int64_t n_rand, uint32_t seed, uint32_t step, int32_t *scratch, uint32_t *indices, float *xyzs,
                     lnerf_stream_t stream);
/* grid[idx] = max(grid[idx] * decay, max of the new_sigmas listed for idx) for every listed cell with a new density
 * >= 0 (cells holding a negative value are never updated).  indices == NULL means cells 0..n-1.  scratch_cells: one
 * uint32 per cell of the level, all zero on entry; left all zero. */
int lnerf_occ_update(float *grid_level, const uint32_t *indices, int64_t n, const float *new_sigmas, float decay,
                     uint32_t *scratch_cells, lnerf_stream_t stream);
/* mean of max(grid,0) over n cells -> *mean_dev ; scratch256: 256 floats of device scratch */
/* lnerf_occ_update + lnerf_occ_mean of ONE cascade level in three launches instead of four, the apply pass streaming over
 * the CELLS (n_cells) instead of exchanging one scratch word per candidate: same grid, same mean, bit for bit.  For a
 * renderer with a single cascade (bound <= 1). */
int lnerf_occ_update_mean(float *grid_level, int64_t n_cells, const uint32_t *indices, int64_t n, const float *new_sigmas,
                          float decay, uint32_t *scratch_cells, float *mean_dev, float *scratch256,
                          lnerf_stream_t stream);
int lnerf_occ_mean(const float *grid, int64_t n, float *mean_dev, float *scratch256, lnerf_stream_t stream);

/* ---- H11: background net, frequency encoding (degree 6: 39 dims) -> 64 -> C, one thread per ray. */
int lnerf_bg_forward(const float *dirs, int64_t N, const float *w1, const float *b1, const float *w2, const float *b2,
                     int C, float *out, lnerf_stream_t stream);
int lnerf_bg_backward(const float *dirs, int64_t N, const float *w1, const float *b1, const float *w2, const float *b2,
                      int C, const float *dout, float *dw1, float *db1, float *dw2, float *db2, lnerf_stream_t stream);

/* ---- sketch-shape guidance (the reference names igl's winding number, README.md:119-122).
 * triangles [F, 3, 3] f32 (vertex positions), points [n,3].  Brute force with LDS-tiled triangles; meant for a
 * one-off evaluation on a dense grid. */
int lnerf_mesh_winding_number(const float *points, int64_t n, const float *triangles, int n_faces, float *out,
                              lnerf_stream_t stream);
int lnerf_mesh_distance(const float *points, int64_t n, const float *triangles, int n_faces, float *out,
                        lnerf_stream_t stream);

/* ---- Latent-Paint raster path (the kaolin calls of src/latent_paint/models/render.py:34-69).
 * cam_host: 14 host floats = world->camera rotation rows (9), camera position (3), fx, fy (= 1/tan(fov/2)).
 * prepare_vertices -> face_z [F,3] (camera z), face_xy [F,3,2] (NDC);  rasterize -> face_idx [H*W] (-1 =
 * background) and perspective-correct barycentrics [H*W,3];  interpolate_attributes: per-face-vertex
 * attributes [F,3,D] -> [H*W,D] (differentiable w.r.t. the attributes);  texture_map: tex [C,R,R] sampled at
 * uv [H*W,2] with grid_sample(align_corners=False, padding 'border') semantics on (u, 1-v), mode 0 nearest /
 * 1 bilinear / 2 bicubic (A = -0.75, every tap clamped to the border) = `guide.texture_interpolation_mode` of
 * src/latent_paint/configs/train_config.py:42-43; pixels with face_idx < 0 give 0.  The backward entry points
 * ACCUMULATE (+=). */
int lnerf_raster_prepare(const float *verts, int n_verts, const int32_t *faces, int n_faces, const float *cam_host,
                         float *face_z, float *face_xy, lnerf_stream_t stream);
int lnerf_rasterize(int H, int W, const float *face_z, const float *face_xy, int n_faces, int32_t *face_idx,
                    float *bary, lnerf_stream_t stream);
int lnerf_interpolate_attributes(const int32_t *face_idx, const float *bary, const float *attr, int n_pixels, int D,
                                 float *feat, lnerf_stream_t stream);
int lnerf_interpolate_attributes_backward(const int32_t *face_idx, const float *bary, const float *dfeat,
                                          int n_pixels, int D, float *dattr, lnerf_stream_t stream);
int lnerf_texture_map_forward(const float *uv, const int32_t *face_idx, const float *texture, int n_pixels, int C,
                              int R, int mode, float *out, lnerf_stream_t stream);
int lnerf_texture_map_backward(const float *uv, const int32_t *face_idx, const float *dout, int n_pixels, int C, int R,
                               int mode, float *dtexture, lnerf_stream_t stream);

/* ---- optimiser step used by the bench/trainer (Adam, src/latent_paint/training/trainer.py:93-95:
 * betas (0.9, 0.99), eps 1e-15).  g is multiplied by grad_scale (1/world_size) first; if
 * zero_grad != 0 the gradient is cleared in the same pass; if shadow_bf16 != NULL the bf16
 * copy read by the gather is refreshed in the same pass.  grad_dtype: LNERF_F32, or LNERF_BF16 to
 * consume the bf16 wire buffer of the data-parallel all-reduce directly. */
int lnerf_adam_step(float *p, void *g, int grad_dtype, float *m, float *v, void *shadow_bf16, int64_t n, float lr,
                    float beta1, float beta2, float eps, int step, const int32_t *step_dev, float grad_scale,
                    int zero_grad, lnerf_stream_t stream);
/* `step_dev` (may be NULL): device-side step counter read by the kernels instead of the host `step`, so a
 * captured hipGraph of the whole optimisation step can be replayed; lnerf_adam_tick() increments it. */
int lnerf_adam_tick(int32_t *step_dev, lnerf_stream_t stream);
/* The same update for up to 16 small tensors in ONE launch (MLP / background parameters).  The pointer
 * and size arrays are HOST arrays of `count` entries holding device pointers.
 * zero_grad | LNERF_ADAM_TICK: the launch also advances *step_dev once all its workgroups have read it (saves the
 * lnerf_adam_tick dispatch when this is the step's last Adam launch); step_dev is then int32[2], [1] = 0 on entry
 * (arrival counter, 0 again on exit). */
#define LNERF_ADAM_TICK 2
int lnerf_adam_step_multi(int count, float *const *p_host, float *const *g_host, float *const *m_host,
                          float *const *v_host, const int64_t *n_host, const float *lr_host, float beta1, float beta2,
                          float eps, int step, const int32_t *step_dev, float grad_scale, int zero_grad,
                          lnerf_stream_t stream);
/* The same launch, which also mirrors tensors into a bf16 image: map_host[k] (NULL: tensor k is not mirrored) holds two
 * int32 positions per element of tensor k (-1: none); the updated value is stored as bf16 at shadow_bf16[position].
 * With the maps of lnerf_mlp_fragment_maps and the head of the MLP workspace as the image, the optimiser keeps the MLP's
 * weight fragments current and lnerf_mlp_forward(... | LNERF_MLP_FRAGMENTS_READY) skips its per-step build (one
 * dispatch of the step). */
int lnerf_adam_step_multi_shadow(int count, float *const *p_host, float *const *g_host, float *const *m_host,
                                 float *const *v_host, const int64_t *n_host, const float *lr_host, float beta1,
                                 float beta2, float eps, int step, const int32_t *step_dev, float grad_scale,
                                 int zero_grad, const int32_t *const *map_host, void *shadow_bf16,
                                 lnerf_stream_t stream);
/* ---- the TAIL of a single-GPU optimisation step: what is left of it besides the scatter (a dependent dispatch in a
 * replayed graph costs ~4.5 us whatever it computes; three of them sat behind the scatter for a few microseconds of work):
 *   - sum of the MLP's gradient slabs (after lnerf_mlp_backward(precision | LNERF_MLP_DEFER_REDUCE)) in the fixed order
 *     of the ordinary reduction, and the Adam step of w1, b1, w2, b2, w3, b3 straight from the sums
 *     (params / exp_avg / exp_avg_sq: host arrays of six device pointers; maps_host: optional, the three fragment maps of
 *     lnerf_mlp_fragment_maps -- the updated weights are mirrored into the fragment image at the head of mlp_workspace;
 *     mlp_workspace = NULL: skipped -- the tick / the clearing below still run, as one workgroup);
 *   - LNERF_TAIL_TICK: *step_dev += 1 once every workgroup has read it (the arrival counters live in the header of the
 *     scatter workspace: LNERF_SCATTER_ZERO_HEAD_BYTES);
 *   - LNERF_TAIL_CLEAR_SCATTER: the scatter's level maxima (first lnerf_grid_scatter_clear_bytes() bytes of its
 *     workspace) are zero on exit, i.e. the NEXT scatter call may pass LNERF_SCATTER_CLEARED.
 * Same arithmetic as the separate launches (csrc/adam_shared.h): parameters and moments are bit-identical.
 * Two forms:
 *   lnerf_grid_encode_backward_adam_tail  the scatter with the fused table update (lnerf_grid_encode_backward_adam) whose
 *     pass 2 ALSO runs the tail, as extra workgroups of the same launch: the step has no launch behind the scatter.
 *     For a step whose only parameters are the table and the MLP's six tensors; needs step_dev and m_host > 0.
 *   lnerf_step_tail  the tail as a launch of its own, behind lnerf_grid_encode_backward_adam (when other small parameters
 *     are stepped in between).  num_levels = 0: no scatter workspace (then no tick / clear). */
#define LNERF_TAIL_TICK 1
#define LNERF_TAIL_CLEAR_SCATTER 2
int lnerf_step_tail(int num_levels, int level_dim, const int32_t *offsets_host, const float *scales_host,
                    const int32_t *res_host, int64_t m_host, int variant, void *scatter_workspace,
                    size_t scatter_workspace_bytes, float *dtable_zero, float *table, float *exp_avg, float *exp_avg_sq,
                    void *shadow_bf16, float table_lr, const void *mlp_workspace, size_t mlp_workspace_bytes,
                    int mlp_precision, int out_dim, float *const *params_host, float *const *exp_avg_host,
                    float *const *exp_avg_sq_host, float mlp_lr, const int32_t *const *maps_host, float beta1, float beta2,
                    float eps, int step, int32_t *step_dev, float grad_scale, int flags, lnerf_stream_t stream);
int lnerf_grid_encode_backward_adam_tail(const float *xyzs, float bound, const void *dfeat, int dfeat_dtype, int num_levels,
                                         int level_dim, const int32_t *offsets_host, const float *scales_host,
                                         const int32_t *res_host, int64_t m_host, const int32_t *m_dev,
                                         int64_t level_stride, float *dtable_zero, int variant, void *workspace,
                                         size_t workspace_bytes, float *table, float *exp_avg, float *exp_avg_sq,
                                         void *shadow_bf16, float lr, const void *mlp_workspace,
                                         size_t mlp_workspace_bytes, int mlp_precision, int out_dim,
                                         float *const *params_host, float *const *exp_avg_host,
                                         float *const *exp_avg_sq_host, float mlp_lr, const int32_t *const *maps_host,
                                         float beta1, float beta2, float eps, int step, int32_t *step_dev, float grad_scale,
                                         int flags, lnerf_stream_t stream);
int lnerf_cast_f32_to_bf16(const float *src, void *dst, int64_t n, lnerf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LNERF_HIP_H */
