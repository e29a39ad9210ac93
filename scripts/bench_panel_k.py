"""Times athena_panel_gemm_k against athena_gemm on the x_net Dense shape (8000 x 5120 -> 256), with the
input rotating through buffers larger than the last-level cache."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from athena_amd import ops
from athena_amd.layers.commons import Dense, bind_all
from athena_amd.runtime import ParamStore
dev = torch.device("cuda")
rows, K = 8000, int(os.environ.get("K", "5120"))
store = ParamStore(torch.bfloat16, device=dev)
dense = Dense(store, "x", K, 256, None, tiled=True)
store.finalize(seed=1)
bind_all(dense)
store.refresh_shadows(force=True)
nbuf = int(os.environ.get("NBUF", "5"))
a = [torch.randn(rows, K, device=dev).to(torch.bfloat16) for _ in range(nbuf)]
out = torch.empty(rows, 256, dtype=torch.bfloat16, device=dev)
pieces = [(dense.tiled_t[8 * c * 512:], K) for c in range(K // 256)]


def panel(x):
    ops.panel_gemm_k(x, rows, K, pieces, out, 256, bias=dense.bias)


def tile(x):
    ops.gemm(x, dense.shadow_t, out, rows, 256, K, trans_b=True, lda=K, ldb=dense.shadow_t.shape[1], ldc=256, bias=dense.bias)


for name, fn in (("panel_k", panel), ("gemm", tile)):
    for x in a:
        fn(x)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(20):
            fn(a[i % nbuf])
    g.replay(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    print("%s K %d buffers %d: %.1f us" % (name, K, nbuf, (time.perf_counter() - t) / 200 * 1e6), flush=True)
