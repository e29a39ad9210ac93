"""conv2 forward / input gradient at batch sizes that give 2, 2.44 (the benchmark) and 3 full rounds of
256x256 tiles on 256 CUs: what the partial last round costs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops
from bench import KernelTimer
dev = torch.device("cuda", 0)
cd = torch.bfloat16
H, W, C, Ho, Wo = 500, 40, 256, 250, 20
for B in (26, 32, 38):
    x = torch.randn(B, H, W, C, device=dev).to(cd)
    w_t = (torch.randn(C, 3, 3, C, device=dev) * 0.05).to(cd)
    w_hwio = (torch.randn(3, 3, C, C, device=dev) * 0.05).to(cd)
    dy = torch.randn(B, Ho, Wo, C, device=dev).to(cd)
    out = torch.empty(B, Ho, Wo, C, device=dev, dtype=cd)
    din = torch.empty(B, H, W, C, device=dev, dtype=cd)
    stats = torch.zeros(2 * C, dtype=torch.float64, device=dev)
    fl = 2.0 * 9 * C * C * B * Ho * Wo
    for _ in range(3):
        ops.conv3x3s2_fwd(x, w_t, 1, out, B, H, W, C, C, stats)
        ops.conv3x3s2_dgrad(dy, w_hwio, din, B, H, W, C, C)
    torch.cuda.synchronize()
    with KernelTimer(["athena_conv3x3s2_fwd", "athena_conv3x3s2_dgrad"]) as kt:
        for _ in range(10):
            ops.conv3x3s2_fwd(x, w_t, 1, out, B, H, W, C, C, stats)
            ops.conv3x3s2_dgrad(dy, w_hwio, din, B, H, W, C, C)
        torch.cuda.synchronize()
    f, c1, _ = kt.total_ms(["athena_conv3x3s2_fwd"])
    d, c2, _ = kt.total_ms(["athena_conv3x3s2_dgrad"])
    tiles = (B * Ho * Wo + 255) // 256
    print("B %d: %d tiles (%.2f rounds)  fwd %.1f us (%.0f TF/s)  dgrad %.1f us (%.0f TF/s)" %
          (B, tiles, tiles / 256.0, 1e3 * f / c1, fl / (1e3 * f / c1) / 1e6, 1e3 * d / c2, fl / (1e3 * d / c2) / 1e6))
    del x, dy, out, din
