import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops
from scripts.bench_kernels import timeit, dev
for M, N, K in [(160000, 256, 2304), (4096, 4096, 4096), (8192, 8192, 1024), (8000, 1024, 64), (8000, 1024, 256), (8000, 1024, 1024), (64000, 1024, 256), (8000, 256, 64), (8000, 256, 256), (8000, 256, 1024), (32000, 256, 256), (1088, 256, 256), (1088, 1024, 256), (1088, 256, 1024), (8000, 4240, 256)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.gemm(a, b, c, M, N, K, trans_b=True), reps=5)
    print("M=%6d N=%5d K=%5d %9.1f us %7.1f TFLOP/s" % (M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12))
