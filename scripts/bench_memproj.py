"""The decoder's memory projections (12 products 8000 x 256 x 256 with bias, bf16) three ways: the grouped
launch of the 128 x 128 core, one product with the twelve kernels stacked (N = 3072) on the large-tile
core, and the same in two and three column parts."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


M, d, n = 8000, 256, 12
x = torch.randn(M, d, device=dev).bfloat16()
w = [torch.randn(d, d, device=dev).bfloat16() * 0.05 for _ in range(n)]
bias = [torch.randn(d, device=dev) for _ in range(n)]
outs = [torch.empty(M, d, device=dev, dtype=torch.bfloat16) for _ in range(n)]
probs = [(x, w[i], outs[i], M, d, d, dict(trans_b=True, lda=d, ldb=d, ldc=d, bias=bias[i])) for i in range(n)]
print("grouped launch, 12 problems: %6.1f us" % timeit(lambda: ops.gemm_grouped(probs)))
probs3 = probs[:3]
print("grouped launch,  3 problems: %6.1f us" % timeit(lambda: ops.gemm_grouped(probs3)))
wall = torch.cat(w, 0).contiguous()
ball = torch.cat(bias, 0).contiguous()
big = torch.empty(M, n * d, device=dev, dtype=torch.bfloat16)
print("one product, N = 3072:       %6.1f us" % timeit(
    lambda: ops.gemm(x, wall, big, M, n * d, d, trans_b=True, lda=d, ldb=d, ldc=n * d, bias=ball)))
ref = torch.cat(outs, 1).float()
print("max difference", float((big.float() - ref).abs().max()))
for parts in (2, 3):
    nn = n * d // parts
    def run():
        for p in range(parts):
            ops.gemm(x, wall[p * nn:(p + 1) * nn], big[:, p * nn:], M, nn, d, trans_b=True, lda=d, ldb=d, ldc=n * d,
                     bias=ball[p * nn:(p + 1) * nn])
    print("%d products, N = %d:        %6.1f us" % (parts, nn, timeit(run)))
w3 = torch.cat(w[:3], 0).contiguous()
b3 = torch.cat(bias[:3], 0).contiguous()
o3 = torch.empty(M, 3 * d, device=dev, dtype=torch.bfloat16)
os.environ["ATHENA_GEMM_BIG"] = "1"
print("one product, N = 768:        %6.1f us" % timeit(
    lambda: ops.gemm(x, w3, o3, M, 3 * d, d, trans_b=True, lda=d, ldb=d, ldc=3 * d, bias=b3)))
