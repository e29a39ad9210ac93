"""Kernel time of athena_seq_errors at the benchmark shape, with and without the table sweep."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops
from bench import KernelTimer
dev = torch.device("cuda", 0)
B, L, U, V = 32, 250, 32, 4233
g = torch.Generator().manual_seed(0)
ids = torch.randint(0, V, (B, L), generator=g, dtype=torch.int32).to(dev)
labels = torch.randint(1, V - 1, (B, U), generator=g, dtype=torch.int32).to(dev)
counts = torch.zeros(2, dtype=torch.int64, device=dev)
for name, lens in (("full hypotheses", torch.full((B,), L, dtype=torch.int32)), ("empty hypotheses", torch.zeros(B, dtype=torch.int32))):
    lens = lens.to(dev)
    for _ in range(3):
        ops.seq_errors(ids, lens, labels, 0, V - 1, counts=counts)
    torch.cuda.synchronize()
    with KernelTimer(["athena_seq_errors"]) as kt:
        for _ in range(20):
            ops.seq_errors(ids, lens, labels, 0, V - 1, counts=counts)
        torch.cuda.synchronize()
    ms, calls, _ = kt.total_ms(["athena_seq_errors"])
    print("%-18s %.1f us" % (name, 1e3 * ms / calls))
