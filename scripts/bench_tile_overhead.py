"""What a tile of the large-tile core costs besides its k-steps: conv2-shaped forward products with 508 tiles
(two full rounds on 256 CUs) at 36, 72 and 144 k-steps per tile, with and without the column statistics;
time per round = overhead + steps x step time.  (Measured outside a recorded graph: the statistics variant
also pays the launch of its zero fill, which shows on the short products only.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops
dev = torch.device("cuda", 0)
cd = torch.bfloat16


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


B, H, W, Ho, Wo, Cout = 26, 500, 40, 250, 20, 256
res = {}
for Cin in (128, 256, 512):
    x = torch.randn(B, H, W, Cin, device=dev).to(cd)
    w_t = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05).to(cd)
    out = torch.empty(B, Ho, Wo, Cout, device=dev, dtype=cd)
    stats = torch.zeros(2 * Cout, dtype=torch.float64, device=dev)
    for st in (None, stats):
        t = timeit(lambda: ops.conv3x3s2_fwd(x, w_t, 1, out, B, H, W, Cin, Cout, st))
        res[(Cin, st is not None)] = t
        print("Cin %3d, %3d k-steps per tile, statistics %-5s: %6.1f us = %5.1f us per round" %
              (Cin, 9 * Cin // 32, st is not None, t, t / 2))
for st in (False, True):
    step = (res[(512, st)] - res[(128, st)]) / 2 / (144 - 36)
    print("statistics %-5s: %.3f us per k-step, %.1f us per tile besides" % (st, step, res[(256, st)] / 2 - 72 * step))
# input gradient: 2500 tiles of 8, 16, 16 and 32 k-steps
B = 32
dy = torch.randn(B, Ho, Wo, 256, device=dev).to(cd)
w_hwio = (torch.randn(3, 3, 256, 256, device=dev) * 0.05).to(cd)
din = torch.empty(B, H, W, 256, device=dev, dtype=cd)
t = timeit(lambda: ops.conv3x3s2_dgrad(dy, w_hwio, din, B, H, W, 256, 256))
print("input gradient, B 32: %.1f us; 45000 k-steps on 256 CUs = 175.8 per CU, 2500 tiles = 9.8 per CU" % t)
