import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops
from scripts.bench_kernels import timeit, dev
d = 256
for rows in (1000, 4000, 8000, 16000, 64000):
    for p in (0.0, 0.1):
        a = torch.randn(rows, d, device=dev).to(torch.bfloat16)
        b = torch.randn(rows, d, device=dev).to(torch.bfloat16)
        g = torch.ones(d, device=dev); be = torch.zeros(d, device=dev)
        rng = torch.tensor([1, 2], dtype=torch.int64, device=dev)
        z, y, mean, rstd = ops.add_layernorm_fwd(a, b, g, be, 1e-8, p, rng, 3)
        dg = torch.empty(d, device=dev); db = torch.empty(d, device=dev)
        t1 = timeit(lambda: ops.add_layernorm_fwd(a, b, g, be, 1e-8, p, rng, 3))
        t2 = timeit(lambda: ops.add_layernorm_bwd(y, z, mean, rstd, g, p, rng, 3, dg, db))
        print("rows %6d p=%.1f  fwd %6.1f us  bwd(+fold) %6.1f us" % (rows, p, t1 * 1e6, t2 * 1e6))
