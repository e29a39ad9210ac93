"""What each fused epilogue option of the FFN GEMM (8000 x 1024 x 256, bf16) costs on its own."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops  # noqa: E402
from scripts.bench_kernels import dev, timeit  # noqa: E402

M, N, K = 8000, 1024, 256
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
b = torch.randn(N, K, device=dev).to(torch.bfloat16)
c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
pre = torch.empty_like(c)
aux = torch.randn(M, N, device=dev).to(torch.bfloat16)
bias = torch.randn(N, device=dev)
rng = torch.tensor([1, 2], dtype=torch.int64, device=dev)
cases = [
    ("plain", dict()),
    ("bias", dict(bias=bias)),
    ("bias+gelu", dict(bias=bias, act=ops.ACT_GELU)),
    ("bias+pre", dict(bias=bias, c_pre=pre)),
    ("bias+drop", dict(bias=bias, drop_p=0.1, rng_state=rng, rng_site=3)),
    ("bias+gelu+pre+drop", dict(bias=bias, act=ops.ACT_GELU, c_pre=pre, drop_p=0.1, rng_state=rng, rng_site=3)),
    ("dgelu(aux)+drop", dict(act=ops.ACT_MUL_DGELU, aux=aux, ldaux=N, drop_p=0.1, rng_state=rng, rng_site=3)),
    ("dgelu(aux)", dict(act=ops.ACT_MUL_DGELU, aux=aux, ldaux=N)),
]
for name, kw in cases:
    t = timeit(lambda: ops.gemm(a, b, c, M, N, K, trans_b=True, ldb=K, ldc=N, **kw))
    print("%-22s %6.1f us" % (name, t * 1e6))
