// Times ctc_lattice_kernel alone on a synthetic emission table (every sample has a path, so the slow
// path and its null operands are never touched) (hipcc -DPROBE_R=2 [-DCTC_PROBE_...]).
#include "../../athena_amd/csrc/loss.hip"
#include <cstdio>
void athena_set_error(const char*, ...) {}
#include <vector>
#ifndef PROBE_R
#define PROBE_R 2
#endif
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 32, T = argc > 2 ? atoi(argv[2]) : 250;
    const int U = argc > 3 ? atoi(argv[3]) : 32;
    constexpr int R = PROBE_R;
    const int Sp = 32 * R;   /* doubles per emission row */
    std::vector<double> e((size_t)B * T * Sp);
    for (size_t i = 0; i < e.size(); ++i) e[i] = 0.001 + 0.01 * ((i * 2654435761u >> 8) % 100);
    std::vector<int> lab((size_t)B * U), ll(B, U), tl(B, T);
    for (size_t i = 0; i < lab.size(); ++i) lab[i] = 1 + (int)((i * 40503u) % 1000);
    double* de; int *dl, *dll, *dtl; double *da, *db, *dz;
    hipMalloc(&de, (e.size() + 32 * Sp) * 8); de += 16 * Sp; hipMalloc(&dl, lab.size() * 4); hipMalloc(&dll, B * 4); hipMalloc(&dtl, B * 4);
    hipMalloc(&da, (size_t)B * T * Sp * 16); hipMalloc(&db, (size_t)B * T * Sp * 16); hipMalloc(&dz, B * 8);
    hipMemcpy(de, e.data(), e.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dl, lab.data(), lab.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dll, ll.data(), B * 4, hipMemcpyHostToDevice);
    hipMemcpy(dtl, tl.data(), B * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; ++i)
            hipLaunchKernelGGL((ctc_lattice_kernel<float, float, R>), dim3(B), dim3(256), 0, 0, de, dl, dll, dtl, T, U, da, db, dz,
                               (const float*)nullptr, 8, 8, 7, U + 1, (const float*)nullptr, (float*)nullptr,
                               (const int32_t*)nullptr, (int32_t*)nullptr, (float*)nullptr, 0.f, (float*)nullptr);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("R=%d B=%d T=%d: %.1f us per launch, %.0f cycles per step at 2.4 GHz\n", R, B, T, ms * 50.f, ms * 50.f * 2400.f / T);
    }
    double z[4]; hipMemcpy(z, dz, 32, hipMemcpyDeviceToHost);
    printf("ll %.4f %.4f (%s)\n", z[0], z[1], hipGetErrorString(hipGetLastError()));
    return 0;
}
