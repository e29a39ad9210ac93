// Latency of dependent VALU operations and issue rate of independent ones, one wave on one SIMD
// (hipcc --offload-arch=gfx950 -O3).  Prints nanoseconds and cycles at 2.4 GHz per operation.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 20000
#define DEP8(stmt) stmt stmt stmt stmt stmt stmt stmt stmt
template <int OP>
__global__ void lat_kernel(double* out, double seed, int sh) {
    double a = seed + threadIdx.x, b = 1.0000001, c = 1e-9;
    float fa = (float)a, fb = 1.0000001f, fc = 1e-9f;
    double a2 = a + 1, a3 = a + 2, a4 = a + 3;
    for (int i = 0; i < N_ITER; ++i) {
        if (OP == 0) { DEP8(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        if (OP == 1) { DEP8(asm volatile("v_add_f64 %0, %0, %1" : "+v"(a) : "v"(c));) }
        if (OP == 2) { DEP8(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (OP == 3) { DEP8(asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(a) : "v"(sh));) }
        if (OP == 4) { DEP8(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fb), "v"(fc));) }
        if (OP == 5) { DEP8(asm volatile("v_exp_f32 %0, %0" : "+v"(fa));) }
        if (OP == 6) { DEP8(asm volatile("v_log_f32 %0, %0" : "+v"(fa));) }
        if (OP == 7) { DEP8(asm volatile("s_nop 1\n v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(fa));) }
        if (OP == 8) { DEP8(asm volatile("v_cvt_f64_f32 %0, %1\n v_cvt_f32_f64 %1, %0" : "+v"(a), "+v"(fa));) }
        // independent streams: issue rate
        if (OP == 9) { DEP8(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(a), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(b), "v"(c));) }
        if (OP == 10) { DEP8(asm volatile("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4" : "+v"(a), "+v"(a2), "+v"(a3), "+v"(a4) : "v"(c));) }
        if (OP == 11) { DEP8(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        if (OP == 12) { DEP8(asm volatile("v_add_f32 %0, %0, %1" : "+v"(fa) : "v"(fc));) }
        if (OP == 13) { DEP8(asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(fa) : "v"(fb), "v"(fc));) }
    }
    out[threadIdx.x] = a + a2 + a3 + a4 + fa;
}
template <int OP>
void run(const char* name, int per_iter) {
    double* d; hipMalloc(&d, 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(lat_kernel<OP>, dim3(1), dim3(64), 0, 0, d, 1.0, 0);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(lat_kernel<OP>, dim3(1), dim3(64), 0, 0, d, 1.0, 0);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)N_ITER * per_iter);
    printf("%-44s %7.2f ns = %6.1f cycles at 2.4 GHz\n", name, ns, ns * 2.4);
    hipFree(d);
}
int main() {
    run<0>("v_fma_f64 dependent", 8); run<1>("v_add_f64 dependent", 8); run<2>("v_mul_f64 dependent", 8);
    run<3>("v_ldexp_f64 dependent", 8); run<4>("v_fma_f32 dependent", 8); run<12>("v_add_f32 dependent", 8);
    run<13>("v_max3_f32 dependent", 8); run<5>("v_exp_f32 dependent", 8);
    run<6>("v_log_f32 dependent", 8); run<7>("s_nop 1 + v_mov_b32_dpp wave_shr dependent", 8);
    run<8>("v_cvt_f64_f32 + v_cvt_f32_f64 pair", 8); run<9>("v_fma_f64 x4 independent (per op)", 32);
    run<10>("v_add_f64 x4 independent (per op)", 32); run<11>("v_pk_fma_f32 dependent", 8);
    return 0;
}
