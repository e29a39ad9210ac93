"""What the memory system gives a plain streaming kernel at sizes inside and beyond the 256 MB last-level cache:
fill, copy (1 read : 1 write) and an fp32 -> bf16 cast (2 : 1 in bytes, the mix of the CTC row launch), each over
a working set of 0.1, 0.4, 0.8 and 1.6 GB."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


for mb in (96, 384, 768, 1536):
    n = mb * (1 << 20) // 6          # fp32 source + bf16 destination = 6 bytes per element
    src = torch.randn(n, device=dev)
    dst = torch.empty(n, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.cast(src, dst))
    line = "%5d MB: cast fp32 -> bf16 %7.1f us = %5.2f TB/s" % (mb, t * 1e6, 6.0 * n / t / 1e12)
    half = torch.empty(n * 6 // 8, device=dev)
    half2 = torch.empty_like(half)
    t = timeit(lambda: half2.copy_(half))
    line += "   copy %7.1f us = %5.2f TB/s" % (t * 1e6, 8.0 * half.numel() / t / 1e12)
    t = timeit(lambda: half.zero_())
    line += "   fill %7.1f us = %5.2f TB/s" % (t * 1e6, 4.0 * half.numel() / t / 1e12)
    print(line)
    del src, dst, half, half2
# read-only: a sum over fp32 (torch's reduction kernel) and the library's sum of squares
for mb in (96, 384, 768, 1536):
    n = mb * (1 << 20) // 4
    src = torch.randn(n, device=dev)
    t = timeit(lambda: src.sum())
    off = torch.tensor([0, n], dtype=torch.int64, device=dev)
    print("%5d MB: read-only sum %7.1f us = %5.2f TB/s" % (mb, t * 1e6, 4.0 * n / t / 1e12))
    del src
