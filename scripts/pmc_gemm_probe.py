"""Three launches each of the 4096^3 bf16 GEMM and the conv2-shaped implicit GEMM, for a
``rocprofv3 --pmc`` pass over the GEMM core's main loop (run it directly after ``--``)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
M = N = K = 4096
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
b = torch.randn(N, K, device=dev).to(torch.bfloat16)
c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
for _ in range(3):
    ops.gemm(a, b, c, M, N, K, trans_a=False, trans_b=True, ldb=K, ldc=N)
B, H, W, C = 32, 500, 40, 256
x = torch.randn(B, H, W, C, device=dev).to(torch.bfloat16)
wt = (torch.randn(C, 3, 3, C, device=dev) * 0.05).to(torch.bfloat16)
out = torch.empty(B, 250, 20, C, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.conv3x3s2_fwd(x, wt, 1, out, B, H, W, C, C, None)
torch.cuda.synchronize()
print("done")
