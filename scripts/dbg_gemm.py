import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops
from scripts.bench_kernels import timeit, dev
for M, N, K in [(8000, 1024, 64), (8000, 1024, 256), (8000, 1024, 1024), (8000, 256, 256)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.gemm(a, b, c, M, N, K, trans_b=True), reps=20)
    print("dbg=%s M=%6d N=%5d K=%5d %9.1f us" % (os.environ.get("ATHENA_GEMM_DEBUG"), M, N, K, t * 1e6))
