"""The vocabulary projections and the other plain products of the step that run on the 128x128 core,
on both cores (ATHENA_GEMM_BIG_MIN_K / ATHENA_GEMM_BIG decide): time and result against fp64."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)


def run(M, N, K, out_dtype, bias, trans_b=True):
    a = torch.randn(M, K, device=dev).bfloat16()
    ldb = (K + 7) // 8 * 8
    b = torch.randn(N, ldb, device=dev).bfloat16() * 0.1
    bi = torch.randn(N, device=dev) if bias else None
    ldc = (N + 7) // 8 * 8
    c = torch.empty(M, ldc, dtype=out_dtype, device=dev)
    f = lambda: ops.gemm(a, b, c, M, N, K, trans_b=True, lda=K, ldb=ldb, ldc=ldc, bias=bi)  # noqa: E731
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    ref = a[:256].double() @ b[:, :K].double().t() + (bi.double() if bias else 0)
    err = float((c[:256, :N].double() - ref).abs().max() / ref.abs().max())
    print("M %5d N %5d K %5d %s bias %d: %7.1f us  %6.1f TFLOP/s  rel err %.1e" %
          (M, N, K, str(out_dtype).split(".")[1], bias, t, 2.0 * M * N * K / t / 1e6, err))


run(8000, 4233, 256, torch.float32, True)
run(8000, 4233, 256, torch.bfloat16, True)
run(1056, 4233, 256, torch.float32, True)
run(8000, 2048, 256, torch.bfloat16, True)
