"""Times athena_panel_gemm on the benchmark's shapes (ROWS rows, default 8000; 3, 12 and 20 blocks), 20 launches
back to back in a replayed graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from athena_amd import ops
dev = torch.device("cuda")
rows = int(os.environ.get("ROWS", "8000"))
x = torch.randn(rows, 256, device=dev).to(torch.bfloat16)
w = torch.randn(20, 65536, device=dev).to(torch.bfloat16)
b = torch.randn(20, 256, device=dev)
for nb in (3, 12, 20):
    out = torch.empty(rows, 256 * nb, dtype=torch.bfloat16, device=dev)
    blocks = [(w[c], b[c], out[:, 256 * c:], 256 * nb) for c in range(nb)]
    for _ in range(5):
        ops.panel_gemm(x, rows, blocks)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            ops.panel_gemm(x, rows, blocks)
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print("panel_gemm rows %d blocks %2d: %.1f us" % (rows, nb, (time.perf_counter() - t) / 200 * 1e6), flush=True)
