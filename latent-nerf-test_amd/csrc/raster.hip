// Latent-Paint raster path (BASELINE config 5): the kaolin ops the
// reference calls in src/latent_paint/models/render.py:34-69 (prepare_vertices :39-40,56-57, rasterize
// :42-43,59-60, texture_mapping :64), rebuilt as HIP kernels.  kaolin itself is not available (un-pinned git
// HEAD, setup.sh:3), so the semantics are the documented ones restated in oracle/raster_oracle.py:
// hard z-buffer (largest camera-space z = closest), perspective-correct barycentric interpolation of
// per-face-vertex attributes, face_idx = -1 on background, texture lookup = grid_sample(align_corners=False,
// padding 'border') on (u, 1 - v).
#include "common.h"

namespace lnerf {

struct Cam {
    float rot[9];   // rows: camera x, y, z axes in world space
    float pos[3];
    float fx, fy;   // 1 / tan(fov/2) (/ ratio)
};

__global__ void __launch_bounds__(256)
k_raster_prepare(const float *__restrict__ verts, const int32_t *__restrict__ faces, int F, Cam cam,
                 float *__restrict__ face_z, float *__restrict__ face_xy) {
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int v = faces[f * 3 + k];
        const float x = verts[v * 3] - cam.pos[0], y = verts[v * 3 + 1] - cam.pos[1], z = verts[v * 3 + 2] - cam.pos[2];
        const float cx = fmaf(cam.rot[0], x, fmaf(cam.rot[1], y, cam.rot[2] * z));
        const float cy = fmaf(cam.rot[3], x, fmaf(cam.rot[4], y, cam.rot[5] * z));
        const float cz = fmaf(cam.rot[6], x, fmaf(cam.rot[7], y, cam.rot[8] * z));
        face_z[f * 3 + k] = cz;
        // image = (x * fx, y * fy) / (z * -1)
        face_xy[(f * 3 + k) * 2] = cx * cam.fx / (-cz);
        face_xy[(f * 3 + k) * 2 + 1] = cy * cam.fy / (-cz);
    }
}

constexpr int RTILE = 128;  // faces per LDS tile

// one thread per pixel; faces stream through LDS
__global__ void __launch_bounds__(256)
k_rasterize(int H, int W, const float *__restrict__ face_z, const float *__restrict__ face_xy, int F,
            int32_t *__restrict__ face_idx, float *__restrict__ bary) {
    __shared__ float s_xy[RTILE * 6];
    __shared__ float s_z[RTILE * 3];
    const int p = blockIdx.x * 256 + threadIdx.x;
    const bool in = p < H * W;
    const int i = in ? p / W : 0, j = in ? p - i * W : 0;
    const float px = (2.0f * (float)j + 1.0f) / (float)W - 1.0f;
    const float py = 1.0f - (2.0f * (float)i + 1.0f) / (float)H;
    float best_z = -3.0e38f;
    int best_f = -1;
    float bw0 = 0.f, bw1 = 0.f, bw2 = 0.f;
    for (int f0 = 0; f0 < F; f0 += RTILE) {
        const int nf = min(RTILE, F - f0);
        __syncthreads();
        for (int k = threadIdx.x; k < nf * 6; k += 256) s_xy[k] = face_xy[(int64_t)f0 * 6 + k];
        for (int k = threadIdx.x; k < nf * 3; k += 256) s_z[k] = face_z[(int64_t)f0 * 3 + k];
        __syncthreads();
        for (int f = 0; f < nf; ++f) {
            const float x0 = s_xy[f * 6], y0 = s_xy[f * 6 + 1], x1 = s_xy[f * 6 + 2], y1 = s_xy[f * 6 + 3];
            const float x2 = s_xy[f * 6 + 4], y2 = s_xy[f * 6 + 5];
            const float z0 = s_z[f * 3], z1 = s_z[f * 3 + 1], z2 = s_z[f * 3 + 2];
            if (!(z0 < 0.f && z1 < 0.f && z2 < 0.f)) continue;  // behind the camera
            const float area = (x1 - x0) * (y2 - y0) - (x2 - x0) * (y1 - y0);
            if (area == 0.f) continue;
            const float e0 = (x1 - px) * (y2 - py) - (x2 - px) * (y1 - py);  // weight of vertex 0
            const float e1 = (x2 - px) * (y0 - py) - (x0 - px) * (y2 - py);  // weight of vertex 1
            const float inv = 1.0f / area;
            const float w0 = e0 * inv, w1 = e1 * inv, w2 = 1.0f - w0 - w1;
            if (w0 < 0.f || w1 < 0.f || w2 < 0.f) continue;
            // perspective-correct weights and depth
            const float q0 = w0 / z0, q1 = w1 / z1, q2 = w2 / z2;
            const float qs = q0 + q1 + q2;
            const float z = 1.0f / qs;   // (sum w_k / z_k)^-1
            if (z > best_z) {            // strictly closer; ties keep the lower face index
                best_z = z;
                best_f = f0 + f;
                bw0 = q0 * z; bw1 = q1 * z; bw2 = q2 * z;
            }
        }
    }
    if (in) {
        face_idx[p] = best_f;
        bary[p * 3] = bw0; bary[p * 3 + 1] = bw1; bary[p * 3 + 2] = bw2;
    }
}

// feat[p, :] = sum_k bary[p,k] * attr[face_idx[p], k, :]   (0 on background)
__global__ void __launch_bounds__(256)
k_interp_attr(const int32_t *__restrict__ face_idx, const float *__restrict__ bary, const float *__restrict__ attr,
              int P, int D, float *__restrict__ feat) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= P * D) return;
    const int p = t / D, d = t - p * D;
    const int f = face_idx[p];
    float v = 0.f;
    if (f >= 0) {
        const float *a = attr + (int64_t)f * 3 * D;
        v = fmaf(bary[p * 3], a[d], fmaf(bary[p * 3 + 1], a[D + d], bary[p * 3 + 2] * a[2 * D + d]));
    }
    feat[t] = v;
}
__global__ void __launch_bounds__(256)
k_interp_attr_bwd(const int32_t *__restrict__ face_idx, const float *__restrict__ bary,
                  const float *__restrict__ dfeat, int P, int D, float *__restrict__ dattr) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= P * D) return;
    const int p = t / D, d = t - p * D;
    const int f = face_idx[p];
    if (f < 0) return;
    const float g = dfeat[t];
    float *a = dattr + (int64_t)f * 3 * D;
    atomicAdd(&a[d], bary[p * 3] * g);
    atomicAdd(&a[D + d], bary[p * 3 + 1] * g);
    atomicAdd(&a[2 * D + d], bary[p * 3 + 2] * g);
}

// texture lookup: tex [C,R,R]; uv [P,2]; mode 0 nearest, 1 bilinear, 2 bicubic; grid_sample(align_corners=False, border)
// on (u, 1 - v) after clamping uv to [0, 1] (the semantics of kal.render.mesh.texture_mapping, the op the reference
// calls at src/latent_paint/models/render.py:64 with mode = guide.texture_interpolation_mode)
__device__ __forceinline__ void tex_coords(float u, float v, int R, bool clip, float &x, float &y) {
    u = clampf(u, 0.f, 1.f);
    v = clampf(v, 0.f, 1.f);
    const float gx = u * 2.0f - 1.0f, gy = -(v * 2.0f - 1.0f);
    x = ((gx + 1.0f) * (float)R - 1.0f) * 0.5f;
    y = ((gy + 1.0f) * (float)R - 1.0f) * 0.5f;
    if (clip) {  // padding_mode = 'border' (nearest / bilinear clip the position, bicubic clips every tap instead)
        x = clampf(x, 0.f, (float)(R - 1));
        y = clampf(y, 0.f, (float)(R - 1));
    }
}
// cubic convolution weights of the 4 taps around a position with fractional part t (A = -0.75, as grid_sample)
__device__ __forceinline__ void cubic_weights(float t, float w[4]) {
    const float A = -0.75f;
    const float a = t + 1.0f, b = 1.0f - t, c = b + 1.0f;
    w[0] = ((A * a - 5.0f * A) * a + 8.0f * A) * a - 4.0f * A;
    w[1] = ((A + 2.0f) * t - (A + 3.0f)) * t * t + 1.0f;
    w[2] = ((A + 2.0f) * b - (A + 3.0f)) * b * b + 1.0f;
    w[3] = ((A * c - 5.0f * A) * c + 8.0f * A) * c - 4.0f * A;
}
template <bool BWD>
__global__ void __launch_bounds__(256)
k_texture_map(const float *__restrict__ uv, const int32_t *__restrict__ face_idx, float *tex, int P, int C, int R,
              int mode, float *out_or_dout) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= P * C) return;
    const int p = t / C, c = t - p * C;
    const bool fg = !face_idx || face_idx[p] >= 0;
    if (!fg) {
        if (!BWD) out_or_dout[t] = 0.f;
        return;
    }
    float x, y;
    tex_coords(uv[p * 2], uv[p * 2 + 1], R, mode != 2, x, y);
    float *tc = tex + (int64_t)c * R * R;
    if (mode == 0) {
        const int xi = (int)rintf(x), yi = (int)rintf(y);
        if (BWD) atomicAdd(&tc[yi * R + xi], out_or_dout[t]);
        else out_or_dout[t] = tc[yi * R + xi];
    } else if (mode == 1) {
        const float xf = floorf(x), yf = floorf(y);
        const int x0 = (int)xf, y0 = (int)yf, x1 = min(x0 + 1, R - 1), y1 = min(y0 + 1, R - 1);
        const float ax = x - xf, ay = y - yf;
        const float w00 = (1.f - ax) * (1.f - ay), w01 = ax * (1.f - ay), w10 = (1.f - ax) * ay, w11 = ax * ay;
        if (BWD) {
            const float g = out_or_dout[t];
            atomicAdd(&tc[y0 * R + x0], w00 * g); atomicAdd(&tc[y0 * R + x1], w01 * g);
            atomicAdd(&tc[y1 * R + x0], w10 * g); atomicAdd(&tc[y1 * R + x1], w11 * g);
        } else {
            out_or_dout[t] = w00 * tc[y0 * R + x0] + w01 * tc[y0 * R + x1] + w10 * tc[y1 * R + x0] + w11 * tc[y1 * R + x1];
        }
    } else {
        const float xf = floorf(x), yf = floorf(y);
        const int xb = (int)xf - 1, yb = (int)yf - 1;
        float wx[4], wy[4];
        cubic_weights(x - xf, wx);
        cubic_weights(y - yf, wy);
        if (BWD) {
            const float g = out_or_dout[t];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int yi = min(max(yb + i, 0), R - 1);
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(&tc[yi * R + min(max(xb + j, 0), R - 1)], (wy[i] * wx[j]) * g);
            }
        } else {
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int yi = min(max(yb + i, 0), R - 1);
                float row = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) row = fmaf(wx[j], tc[yi * R + min(max(xb + j, 0), R - 1)], row);
                acc = fmaf(wy[i], row, acc);
            }
            out_or_dout[t] = acc;
        }
    }
}

}  // namespace lnerf

using namespace lnerf;

extern "C" {

int lnerf_raster_prepare(const float *verts, int n_verts, const int32_t *faces, int n_faces, const float *cam_host,
                         float *face_z, float *face_xy, lnerf_stream_t stream) {
    LNERF_REQUIRE(n_verts > 0 && n_faces > 0, "raster_prepare: empty mesh");
    LNERF_REQUIRE(verts && faces && cam_host && face_z && face_xy, "raster_prepare: null pointer");
    Cam cam;
    for (int i = 0; i < 9; ++i) cam.rot[i] = cam_host[i];
    for (int i = 0; i < 3; ++i) cam.pos[i] = cam_host[9 + i];
    cam.fx = cam_host[12];
    cam.fy = cam_host[13];
    hipLaunchKernelGGL(k_raster_prepare, dim3((unsigned)div_up(n_faces, 256)), dim3(256), 0, as_stream(stream), verts,
                       faces, n_faces, cam, face_z, face_xy);
    LNERF_CHECK_LAUNCH("raster_prepare");
    return LNERF_OK;
}

int lnerf_rasterize(int H, int W, const float *face_z, const float *face_xy, int n_faces, int32_t *face_idx,
                    float *bary, lnerf_stream_t stream) {
    LNERF_REQUIRE(H > 0 && W > 0 && n_faces > 0, "rasterize: bad sizes");
    LNERF_REQUIRE(face_z && face_xy && face_idx && bary, "rasterize: null pointer");
    hipLaunchKernelGGL(k_rasterize, dim3((unsigned)div_up((int64_t)H * W, 256)), dim3(256), 0, as_stream(stream), H, W,
                       face_z, face_xy, n_faces, face_idx, bary);
    LNERF_CHECK_LAUNCH("rasterize");
    return LNERF_OK;
}

int lnerf_interpolate_attributes(const int32_t *face_idx, const float *bary, const float *attr, int n_pixels, int D,
                                 float *feat, lnerf_stream_t stream) {
    LNERF_REQUIRE(n_pixels > 0 && D >= 1 && D <= 16, "interpolate_attributes: bad sizes");
    LNERF_REQUIRE(face_idx && bary && attr && feat, "interpolate_attributes: null pointer");
    hipLaunchKernelGGL(k_interp_attr, dim3((unsigned)div_up((int64_t)n_pixels * D, 256)), dim3(256), 0,
                       as_stream(stream), face_idx, bary, attr, n_pixels, D, feat);
    LNERF_CHECK_LAUNCH("interpolate_attributes");
    return LNERF_OK;
}

int lnerf_interpolate_attributes_backward(const int32_t *face_idx, const float *bary, const float *dfeat,
                                          int n_pixels, int D, float *dattr, lnerf_stream_t stream) {
    LNERF_REQUIRE(n_pixels > 0 && D >= 1 && D <= 16, "interpolate_attributes_backward: bad sizes");
    LNERF_REQUIRE(face_idx && bary && dfeat && dattr, "interpolate_attributes_backward: null pointer");
    hipLaunchKernelGGL(k_interp_attr_bwd, dim3((unsigned)div_up((int64_t)n_pixels * D, 256)), dim3(256), 0,
                       as_stream(stream), face_idx, bary, dfeat, n_pixels, D, dattr);
    LNERF_CHECK_LAUNCH("interpolate_attributes_backward");
    return LNERF_OK;
}

int lnerf_texture_map_forward(const float *uv, const int32_t *face_idx, const float *texture, int n_pixels, int C,
                              int R, int mode, float *out, lnerf_stream_t stream) {
    LNERF_REQUIRE(n_pixels > 0 && C >= 1 && R >= 1 && mode >= 0 && mode <= 2, "texture_map_forward: bad arguments");
    LNERF_REQUIRE(uv && texture && out, "texture_map_forward: null pointer");
    hipLaunchKernelGGL(k_texture_map<false>, dim3((unsigned)div_up((int64_t)n_pixels * C, 256)), dim3(256), 0,
                       as_stream(stream), uv, face_idx, const_cast<float *>(texture), n_pixels, C, R, mode, out);
    LNERF_CHECK_LAUNCH("texture_map_forward");
    return LNERF_OK;
}

int lnerf_texture_map_backward(const float *uv, const int32_t *face_idx, const float *dout, int n_pixels, int C, int R,
                               int mode, float *dtexture, lnerf_stream_t stream) {
    LNERF_REQUIRE(n_pixels > 0 && C >= 1 && R >= 1 && mode >= 0 && mode <= 2, "texture_map_backward: bad arguments");
    LNERF_REQUIRE(uv && dout && dtexture, "texture_map_backward: null pointer");
    hipLaunchKernelGGL(k_texture_map<true>, dim3((unsigned)div_up((int64_t)n_pixels * C, 256)), dim3(256), 0,
                       as_stream(stream), uv, face_idx, dtexture, n_pixels, C, R, mode, const_cast<float *>(dout));
    LNERF_CHECK_LAUNCH("texture_map_backward");
    return LNERF_OK;
}

}  // extern "C"
