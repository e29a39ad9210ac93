"""Time the Dense products of one AISHELL-sized transformer layer in isolation (forward and input-gradient shapes)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops
from scripts.bench_kernels import timeit, dev
M = 8000
for name, N, K in [("qkv fwd", 768, 256), ("o fwd", 256, 256), ("ffn1 fwd", 1024, 256), ("ffn2 fwd", 256, 1024),
                   ("vocab fwd", 4240, 256), ("vocab dgrad", 256, 4240)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.gemm(a, b, c, M, N, K, trans_b=True), reps=20)
    print("%-12s M=%6d N=%5d K=%5d %9.1f us %7.1f TFLOP/s" % (name, M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12))
