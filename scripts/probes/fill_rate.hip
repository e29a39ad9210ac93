// Probe: how fast can one workgroup per CU pull a SHARED, L2-resident block of memory into registers?
// Every workgroup reads the same ``bytes`` (like the weights of a fused layer kernel), 16 bytes per
// lane per load, DEPTH loads in flight per wave.  Prints GB/s per CU for a few depths and wave counts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int DEPTH>
__global__ __launch_bounds__(512) void pull(const f4* __restrict__ src, long n_vec, int shared, float* sink) {
    const int nthreads = blockDim.x;
    const f4* base = src + (shared ? 0 : (long)blockIdx.x * n_vec);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    long i = threadIdx.x;
    for (; i + (long)(DEPTH - 1) * nthreads < n_vec; i += (long)DEPTH * nthreads) {
        f4 v[DEPTH];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) v[d] = base[i + (long)d * nthreads];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <int DEPTH>
static void run(const f4* src, long bytes, int threads, int shared, int grid, float* sink) {
    const long n_vec = bytes / 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(pull<DEPTH>, dim3(grid), dim3(threads), 0, 0, src, n_vec, shared, sink);
    hipEventRecord(e0, 0);
    const int reps = 20;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(pull<DEPTH>, dim3(grid), dim3(threads), 0, 0, src, n_vec, shared, sink);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("bytes/WG %7ld KB  threads %3d  depth %2d  %s  grid %d: %7.1f us  -> %6.1f GB/s per WG, %6.2f TB/s total\n",
           bytes >> 10, threads, DEPTH, shared ? "shared " : "private", grid, us, bytes / us / 1e3, bytes / us / 1e6 * grid);
}

int main() {
    const int grid = 250;
    const long max_bytes = 1200L << 10;
    f4* src;
    float* sink;
    hipMalloc(&src, max_bytes * grid);
    hipMalloc(&sink, 4);
    hipMemset(src, 0, max_bytes * grid);
    for (int shared = 1; shared >= 0; --shared) {
        for (int threads : {256, 512, 1024}) {
            run<4>(src, max_bytes, threads, shared, grid, sink);
            run<8>(src, max_bytes, threads, shared, grid, sink);
            run<16>(src, max_bytes, threads, shared, grid, sink);
        }
    }
    // fewer workgroups: is it the CU or the L2 that limits?
    run<16>(src, max_bytes, 512, 1, 32, sink);
    run<16>(src, max_bytes, 512, 1, 64, sink);
    run<16>(src, max_bytes, 512, 1, 128, sink);
    run<16>(src, max_bytes, 512, 1, 500, sink);
    hipDeviceSynchronize();
    return 0;
}
