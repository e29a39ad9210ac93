import torch, sys, os
sys.path.insert(0, "/root/repo")
from scripts.bench_kernels import timeit, dev
n = 32*500*40*256
x = torch.empty(n, dtype=torch.bfloat16, device=dev)
y = torch.empty(n, dtype=torch.bfloat16, device=dev)
t = timeit(lambda: x.zero_(), reps=5); print("fill 328MB %.1f us %.0f GB/s" % (t*1e6, n*2/t/1e9))
t = timeit(lambda: y.copy_(x), reps=5); print("copy 328MB %.1f us %.0f GB/s (r+w)" % (t*1e6, 2*n*2/t/1e9))
t = timeit(lambda: torch.relu_(x), reps=5); print("relu_ 328MB %.1f us %.0f GB/s (r+w)" % (t*1e6, 2*n*2/t/1e9))
s = torch.empty(n//8, dtype=torch.bfloat16, device=dev)
t = timeit(lambda: s.zero_(), reps=5); print("fill 41MB %.1f us %.0f GB/s" % (t*1e6, n//8*2/t/1e9))
