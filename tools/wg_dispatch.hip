// Microbenchmark (diagnostic, not part of the library): what does a workgroup COST to place on gfx950?  Kernels that do
// (almost) nothing, in the shapes of the scatter's reduce pass: N workgroups of T threads with L bytes of LDS.
//   hipcc -O3 --offload-arch=gfx950 tools/wg_dispatch.hip -o tools/bin/wg_dispatch && tools/bin/wg_dispatch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// WORK = 0: one LDS store per thread and exit.  WORK = 1: also zero LDS bytes [0, 64 KiB) + a barrier (the reduce pass's
// start).  WORK = 2: also stream `bytes_per_wg` from memory (16 B per lane) -- a stand-in for the Adam phase
template <int T, int WORK>
__global__ void __launch_bounds__(T) k(int lds_words, const uint4 *__restrict__ src, uint4 *__restrict__ dst, int per_wg16) {
    extern __shared__ unsigned int lds[];
    if (WORK >= 1) {
        for (int i = threadIdx.x; i < 16384; i += T) lds[i] = 0u;
        __syncthreads();
    } else {
        lds[threadIdx.x] = threadIdx.x;
    }
    if (WORK == 2) {
        const uint4 *s = src + (size_t)blockIdx.x * per_wg16;
        uint4 *d = dst + (size_t)blockIdx.x * per_wg16;
        for (int i = threadIdx.x; i < per_wg16; i += T) {
            typedef unsigned int u4 __attribute__((ext_vector_type(4)));
            u4 v = __builtin_nontemporal_load(reinterpret_cast<const u4 *>(s + i));
            v.x += lds[(threadIdx.x * 7) & 16383];
            __builtin_nontemporal_store(v, reinterpret_cast<u4 *>(d + i));
        }
    }
    if (lds_words < 0) dst[0] = make_uint4(lds[0], 0, 0, 0);
}

template <int T, int WORK>
static void run(const char *name, int n_wg, int lds_bytes, const uint4 *src, uint4 *dst, int per_wg16) {
    CHECK(hipFuncSetAttribute((const void *)k<T, WORK>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<T, WORK>), dim3(n_wg), dim3(T), lds_bytes, 0, lds_bytes / 4, src, dst, per_wg16);
    CHECK(hipDeviceSynchronize());
    const int reps = 20;
    CHECK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k<T, WORK>), dim3(n_wg), dim3(T), lds_bytes, 0, lds_bytes / 4, src, dst, per_wg16);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms = 0.f;
    CHECK(hipEventElapsedTime(&ms, a, b));
    printf("%-58s wgs %5d threads %4d lds %6d B : %7.2f us per launch\n", name, n_wg, T, lds_bytes, ms * 1e3f / reps);
}

int main() {
    const size_t bytes = (size_t)1463 * 224 * 1024;   // 224 KiB per workgroup ~ the Adam phase's p, m, v in and out
    uint4 *src, *dst;
    CHECK(hipMalloc(&src, bytes)); CHECK(hipMalloc(&dst, bytes));
    CHECK(hipMemset(src, 1, bytes)); CHECK(hipMemset(dst, 0, bytes));
    run<1024, 0>("empty, reduce-pass shape", 1463, 77888, src, dst, 0);
    run<1024, 0>("empty, 1024 threads, small LDS", 1463, 4096, src, dst, 0);
    run<512, 0>("empty, 512 threads, half the LDS", 2926, 38944, src, dst, 0);
    run<256, 0>("empty, 256 threads", 5852, 19472, src, dst, 0);
    run<1024, 1>("zero 64 KiB of LDS + barrier", 1463, 77888, src, dst, 0);
    run<1024, 2>("+ stream 112 KiB in, 112 KiB out per workgroup", 1463, 77888, src, dst, 7168);
    run<512, 2>("same bytes, 512 threads (2926 workgroups, 56 + 56 KiB each)", 2926, 38944, src, dst, 3584);
    run<256, 2>("same bytes, 256 threads, small LDS (5852 workgroups)", 5852, 19472, src, dst, 1792);
    run<256, 2>("same bytes, 256 threads, 64 KiB LDS (2 per CU)", 5852, 65536, src, dst, 1792);
    return 0;
}
