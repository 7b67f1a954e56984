// Microbenchmark (diagnostic, not part of the library): LDS atomic / plain read-modify-write rates on gfx950, in the
// shape of the scatter's reduce pass (1024-thread workgroups, two per CU, 4096-row tiles).
//   hipcc -O3 --offload-arch=gfx950 tools/lds_atomics.hip -o tools/bin/lds_atomics && tools/bin/lds_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int ROWS = 4096;

template <int MODE>
__global__ void __launch_bounds__(1024) k(int iters, unsigned long long *sink) {
    __shared__ unsigned long long acc[ROWS * 2];
    unsigned int *acc32 = reinterpret_cast<unsigned int *>(acc);
    for (int i = threadIdx.x; i < ROWS * 2; i += 1024) acc[i] = 0ull;
    __syncthreads();
    unsigned int s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    unsigned long long keep = 0ull;
    const int lane = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            s = s * 1664525u + 1013904223u;
            const unsigned int rnd = (s >> 12) & (ROWS - 1);
            const unsigned int seq = (unsigned int)(lane + (it * 4 + u) * 64) & (ROWS - 1);
            if (MODE == 0) {          // two 64-bit atomics per record, random rows, [feature][row] planes
                atomicAdd(&acc[rnd], (unsigned long long)s);
                atomicAdd(&acc[rnd + ROWS], (unsigned long long)(s >> 3));
            } else if (MODE == 1) {   // one returning 32-bit atomic per record, random
                keep += atomicAdd(&acc32[rnd], 1u);
            } else if (MODE == 2) {   // one non-returning 32-bit atomic per record, random
                atomicAdd(&acc32[rnd], 1u);
            } else if (MODE == 3) {   // two 64-bit atomics, conflict-free consecutive rows
                atomicAdd(&acc[seq], (unsigned long long)s);
                atomicAdd(&acc[seq + ROWS], (unsigned long long)(s >> 3));
            } else if (MODE == 4) {   // plain (non-atomic) 64-bit read-modify-write x2, random
                acc[rnd] += (unsigned long long)s;
                acc[rnd + ROWS] += (unsigned long long)(s >> 3);
            } else if (MODE == 5) {   // returning 32-bit atomic, conflict-free
                keep += atomicAdd(&acc32[seq], 1u);
            } else if (MODE == 6) {   // ONE 64-bit atomic per record, random
                atomicAdd(&acc[rnd], (unsigned long long)s);
            } else if (MODE == 7) {   // two 64-bit atomics on adjacent words ([row][feature])
                atomicAdd(&acc[rnd * 2], (unsigned long long)s);
                atomicAdd(&acc[rnd * 2 + 1], (unsigned long long)(s >> 3));
            } else if (MODE == 8) {   // plain 64-bit store x2 random (no read)
                acc[rnd] = (unsigned long long)s;
                acc[rnd + ROWS] = (unsigned long long)(s >> 3);
            } else if (MODE == 9) {   // returning 32-bit atomic + plain 8-byte store at a rank-dependent slot (counting sort step)
                const unsigned int r = atomicAdd(&acc32[rnd], 1u);
                acc[ROWS + ((rnd + r) & (ROWS - 1))] = (unsigned long long)s;
            }
        }
    }
    __syncthreads();
    if (keep == 0x1234567ull || acc[threadIdx.x] == 0x7777ull) sink[0] = keep;
}

template <int MODE> void run(const char *name, int recs_per_op, unsigned long long *sink) {
    const int iters = 256, grid = 512;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(1024), 0, 0, 8, sink);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(1024), 0, 0, iters, sink);
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    const double recs = (double)grid * 1024 * iters * 4;
    const double ns_per_rec_cu = best * 1e6 / (recs / 256.0);
    printf("%-70s %8.3f ms  %7.3f ns/record/CU  (%.2f records/clk/CU at 2.4 GHz)\n", name, best, ns_per_rec_cu,
           1.0 / (ns_per_rec_cu * 2.4));
}

int main() {
    unsigned long long *sink;
    CHECK(hipMalloc(&sink, 64));
    run<0>("0: 2 x ds_add_u64, random rows, planes (reduce pass today)", 1, sink);
    run<7>("7: 2 x ds_add_u64, random rows, adjacent words", 1, sink);
    run<3>("3: 2 x ds_add_u64, consecutive rows (conflict-free)", 1, sink);
    run<6>("6: 1 x ds_add_u64, random rows", 1, sink);
    run<1>("1: 1 x ds_add_rtn_u32, random", 1, sink);
    run<2>("2: 1 x ds_add_u32 (no return), random", 1, sink);
    run<5>("5: 1 x ds_add_rtn_u32, consecutive", 1, sink);
    run<4>("4: 2 x plain 64-bit read-modify-write, random", 1, sink);
    run<8>("8: 2 x plain 64-bit store, random", 1, sink);
    run<9>("9: ds_add_rtn_u32 + dependent 64-bit store (counting-sort step)", 1, sink);
    return 0;
}
