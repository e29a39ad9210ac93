"""Runs one product of the 256x256 core a few times (for rocprofv3 --pmc passes): conv2 forward at
the AISHELL size, or the table of encoder weight gradients."""
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops

what = sys.argv[1] if len(sys.argv) > 1 else "conv"
dev = torch.device("cuda", 0)
cd = torch.bfloat16
if what == "conv":
    B, H, W, C = 32, 500, 40, 256
    x = torch.randn(B, H, W, C, device=dev).to(cd)
    w_t = (torch.randn(C, 3, 3, C, device=dev) * 0.05).to(cd)
    out = torch.empty(B, 250, 20, C, device=dev, dtype=cd)
    for _ in range(5):
        ops.conv3x3s2_fwd(x, w_t, 1, out, B, H, W, C, C, None)
else:
    rows, d, dff = 8000, 256, 1024
    probs = []
    for layer in range(12):
        xs = torch.randn(rows, d, device=dev).to(cd)
        dh = torch.randn(rows, dff, device=dev).to(cd)
        h = torch.randn(rows, dff, device=dev).to(cd)
        df = torch.randn(rows, d, device=dev).to(cd)
        probs.append((xs, dh, rows, d, dff, d, dff, torch.empty(d, dff, device=dev), None))
        probs.append((h, df, rows, dff, d, dff, d, torch.empty(dff, d, device=dev), None))
    for _ in range(5):
        ops.gemm_wgrad_table(probs)
torch.cuda.synchronize()
