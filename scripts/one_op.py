"""Run a single entry point a few times (for rocprofv3 counter passes)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops
dev = torch.device("cuda", 0)
what = sys.argv[1]
if what == "wgrad":
    B, H, W, C = 32, 500, 40, 256
    x = torch.randn(B, H, W, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, 250, 20, C, device=dev).to(torch.bfloat16)
    dw = torch.empty(3, 3, C, C, device=dev)
    for _ in range(2):
        ops.conv3x3s2_wgrad(x, dy, dw, B, H, W, C, C)
elif what == "gemm4k":
    M = N = K = 4096
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(3):
        ops.gemm(a, b, c, M, N, K, trans_b=True)
elif what == "dw":
    M, N, K = 256, 1024, 8000
    a = torch.randn(K, M, device=dev).to(torch.bfloat16)
    b = torch.randn(K, N, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, device=dev)
    for _ in range(3):
        ops.gemm(a, b, c, M, N, K, trans_a=True, trans_b=False, split_k=0)
torch.cuda.synchronize()
