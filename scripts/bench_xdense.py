import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from athena_amd import ops
dev = torch.device("cuda", 0)
torch.manual_seed(0)
M, N, K = 8000, 256, 5120
a = torch.randn(M, K, device=dev).bfloat16()
b = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
bias = torch.randn(N, device=dev)
for split in (0, 1, 2, 4, 8):
    for od in (torch.bfloat16,):
        c = torch.empty(M, N, dtype=od, device=dev)
        try:
            f = lambda: ops.gemm(a, b, c, M, N, K, trans_b=True, lda=K, ldb=K, ldc=N, split_k=split)
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            print("split", split, str(od), "%.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
        except Exception as e:
            print("split", split, "failed:", str(e)[:100])
# backward dX: dy [M, 256] x W [256(out) ... ] -> [M, 5120]
