// Shader clock against the constant 100 MHz wall clock: one wave spinning alone, and the same wave
// while the rest of the chip streams memory (hipcc --offload-arch=gfx950 -O3).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long* out, int iters) {
    double a = 1.0 + threadIdx.x;
    const long long w0 = wall_clock64(), c0 = clock64();
    for (int i = 0; i < iters; ++i) asm volatile("v_add_f64 %0, %0, %0" : "+v"(a));
    const long long w1 = wall_clock64(), c1 = clock64();
    if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; out[2] = (long long)a; }
}
__global__ void stream(const float4* __restrict__ src, float4* __restrict__ dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    long long* d; hipMalloc(&d, 64);
    long long h[3];
    const long n = 1L << 26;                      // 1 GiB each way
    float4 *a, *b; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16);
    hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
    for (int load = 0; load < 2; ++load) {
        if (load) for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(stream, dim3(4096), dim3(256), 0, s2, a, b, n);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 200000);
        hipDeviceSynchronize();
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("%s: %lld wall ticks (100 MHz) = %.1f us, %lld shader clocks -> %.0f MHz; dependent v_add_f64 = %.2f ns = %.2f clocks\n",
               load ? "chip streaming" : "chip idle", h[0], h[0] / 100.0, h[1], h[1] / (h[0] / 100.0),
               h[0] * 10.0 / 200000, (double)h[1] / 200000);
    }
    return 0;
}
