// Sketch-shape guidance: generalised winding number and unsigned distance of points
// against a triangle mesh, brute force with the triangles tiled through LDS.  The reference's README
// names igl for this (README.md:119-122; the code is absent); here it is evaluated ONCE on a dense grid at
// start-up (G^3 points x F triangles), after which per-sample look-ups are trilinear reads.
//
// Winding number: w(p) = 1/(4 pi) sum_f Omega_f(p), Omega by van Oosterom & Strackee:
//   tan(Omega/2) = det[a b c] / (|a||b||c| + (a.b)|c| + (b.c)|a| + (c.a)|b|),  a,b,c = triangle vertices - p.
#include "common.h"

namespace lnerf {

constexpr int MESH_TILE = 256;  // triangles per LDS tile (9 floats each)

struct P3 { float x, y, z; };
__device__ __forceinline__ P3 sub3(P3 a, P3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ float dot3(P3 a, P3 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, a.z * b.z)); }
__device__ __forceinline__ P3 cross3(P3 a, P3 b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

__device__ __forceinline__ float solid_angle(P3 a, P3 b, P3 c) {
    const float la = sqrtf(dot3(a, a)), lb = sqrtf(dot3(b, b)), lc = sqrtf(dot3(c, c));
    const float det = dot3(a, cross3(b, c));
    const float den = la * lb * lc + dot3(a, b) * lc + dot3(b, c) * la + dot3(c, a) * lb;
    return 2.0f * atan2f(det, den);
}

// closest-point distance^2 from p to triangle (a, b, c)  (Ericson, Real-Time Collision Detection 5.1.5)
__device__ __forceinline__ float tri_dist2(P3 p, P3 a, P3 b, P3 c) {
    const P3 ab = sub3(b, a), ac = sub3(c, a), ap = sub3(p, a);
    const float d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    if (d1 <= 0.f && d2 <= 0.f) return dot3(ap, ap);
    const P3 bp = sub3(p, b);
    const float d3 = dot3(ab, bp), d4 = dot3(ac, bp);
    if (d3 >= 0.f && d4 <= d3) return dot3(bp, bp);
    const float vc = d1 * d4 - d3 * d2;
    if (vc <= 0.f && d1 >= 0.f && d3 <= 0.f) {
        const float v = d1 / (d1 - d3);
        const P3 q = {ap.x - v * ab.x, ap.y - v * ab.y, ap.z - v * ab.z};
        return dot3(q, q);
    }
    const P3 cp = sub3(p, c);
    const float d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    if (d6 >= 0.f && d5 <= d6) return dot3(cp, cp);
    const float vb = d5 * d2 - d1 * d6;
    if (vb <= 0.f && d2 >= 0.f && d6 <= 0.f) {
        const float w = d2 / (d2 - d6);
        const P3 q = {ap.x - w * ac.x, ap.y - w * ac.y, ap.z - w * ac.z};
        return dot3(q, q);
    }
    const float va = d3 * d6 - d5 * d4;
    if (va <= 0.f && (d4 - d3) >= 0.f && (d5 - d6) >= 0.f) {
        const float w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        const P3 q = {bp.x - w * (c.x - b.x), bp.y - w * (c.y - b.y), bp.z - w * (c.z - b.z)};
        return dot3(q, q);
    }
    const float denom = 1.0f / (va + vb + vc);
    const float v = vb * denom, w = vc * denom;
    const P3 q = {ap.x - ab.x * v - ac.x * w, ap.y - ab.y * v - ac.y * w, ap.z - ab.z * v - ac.z * w};
    return dot3(q, q);
}

template <bool DIST>
__global__ void __launch_bounds__(256)
k_mesh_query(const float *__restrict__ pts, int64_t n, const float *__restrict__ tris, int F,
             float *__restrict__ out) {
    __shared__ float s_tri[MESH_TILE * 9];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = i < n;
    P3 p = {0.f, 0.f, 0.f};
    if (in) p = {pts[i * 3], pts[i * 3 + 1], pts[i * 3 + 2]};
    float acc = DIST ? 3.0e38f : 0.f;
    for (int f0 = 0; f0 < F; f0 += MESH_TILE) {
        const int nf = min(MESH_TILE, F - f0);
        __syncthreads();
        for (int k = threadIdx.x; k < nf * 9; k += 256) s_tri[k] = tris[(int64_t)f0 * 9 + k];
        __syncthreads();
        for (int f = 0; f < nf; ++f) {
            const P3 a = {s_tri[f * 9], s_tri[f * 9 + 1], s_tri[f * 9 + 2]};
            const P3 b = {s_tri[f * 9 + 3], s_tri[f * 9 + 4], s_tri[f * 9 + 5]};
            const P3 c = {s_tri[f * 9 + 6], s_tri[f * 9 + 7], s_tri[f * 9 + 8]};
            if (DIST) acc = fminf(acc, tri_dist2(p, a, b, c));
            else acc += solid_angle(sub3(a, p), sub3(b, p), sub3(c, p));
        }
    }
    if (in) out[i] = DIST ? sqrtf(acc) : acc * 0.07957747154594767f;  // 1 / (4 pi)
}

}  // namespace lnerf

using namespace lnerf;

extern "C" {

int lnerf_mesh_winding_number(const float *points, int64_t n, const float *triangles, int n_faces, float *out,
                              lnerf_stream_t stream) {
    LNERF_REQUIRE(n >= 0 && n_faces >= 0, "mesh_winding_number: negative size");
    if (n == 0) return LNERF_OK;
    LNERF_REQUIRE(points && out && (n_faces == 0 || triangles), "mesh_winding_number: null pointer");
    hipLaunchKernelGGL(k_mesh_query<false>, dim3((unsigned)div_up(n, 256)), dim3(256), 0, as_stream(stream), points, n,
                       triangles, n_faces, out);
    LNERF_CHECK_LAUNCH("mesh_winding_number");
    return LNERF_OK;
}

int lnerf_mesh_distance(const float *points, int64_t n, const float *triangles, int n_faces, float *out,
                        lnerf_stream_t stream) {
    LNERF_REQUIRE(n >= 0 && n_faces >= 1, "mesh_distance: need at least one triangle");
    if (n == 0) return LNERF_OK;
    LNERF_REQUIRE(points && out && triangles, "mesh_distance: null pointer");
    hipLaunchKernelGGL(k_mesh_query<true>, dim3((unsigned)div_up(n, 256)), dim3(256), 0, as_stream(stream), points, n,
                       triangles, n_faces, out);
    LNERF_CHECK_LAUNCH("mesh_distance");
    return LNERF_OK;
}

}  // extern "C"
