"""athena_embed_bwd at the three workload shapes, with random ids and with half of the tokens on the padding id."""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from athena_amd import ops
dev = torch.device("cuda", 0)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
rng = torch.tensor([1, 2], dtype=torch.int64, device=dev)
for B, L, d, V in [(32, 33, 256, 4234), (64, 201, 256, 8002), (8, 151, 512, 5002)]:
    ids = torch.randint(1, V, (B, L), device=dev, dtype=torch.int32)
    dy = torch.randn(B, L, d, device=dev).bfloat16()
    g = torch.empty(V, d, device=dev)
    t1 = timeit(lambda: ops.embed_bwd(dy, ids, B, L, d, V, 16.0, 0.1, rng, 3, g))
    ids[:, L // 2:] = 0
    ids[:, 0] = V - 1
    t2 = timeit(lambda: ops.embed_bwd(dy, ids, B, L, d, V, 16.0, 0.1, rng, 3, g))
    print("B %d L %d d %d V %d: %.1f us with random ids, %.1f us with half of the tokens on id 0" % (B, L, d, V, t1, t2))
