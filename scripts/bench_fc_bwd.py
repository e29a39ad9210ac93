"""Backward reduction of the fused first front-end layer at the benchmark shape: the gate read off the
stored activation against the gate worked out again (ATHENA_FC_GATE = ring depth, ATHENA_FC_WGS =
workgroups), and the two results against each other."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
B, H, W, C = 32, 1000, 80, 256
Ho, Wo = (H + 1) // 2, (W + 1) // 2
torch.manual_seed(0)
x = torch.randn(B, H, W, device=dev)
w = torch.randn(3, 3, 1, C, device=dev) * 0.3
gamma = torch.rand(C, device=dev) + 0.5
beta = torch.randn(C, device=dev) * 0.5 + 1.0
mom = torch.empty(ops.conv1_moments_len(), dtype=torch.float64, device=dev)
stats = torch.empty(2 * C, dtype=torch.float64, device=dev)
ops.conv1_bn_stats(x, w, B, H, W, C, mom, stats)
mr = torch.empty(2 * C, dtype=torch.float32, device=dev)
ops.bn_finalize(stats, B * Ho * Wo, C, 1e-3, 0.99, mr, None, None)
out = torch.empty(B, Ho, Wo, C, dtype=torch.bfloat16, device=dev)
ops.conv1_bn_relu6_fwd(x, w, mr, None, gamma, beta, 1e-3, False, B, H, W, C, out)
g = torch.randn(B, Ho, Wo, C, device=dev).bfloat16()


def run(with_beta):
    dw, dg, db = (torch.empty(9, C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev))
    f = lambda: ops.conv1_bn_bwd(x, g, w, mom, mr, gamma, B, H, W, C, dw, dg, db, relu6_out=out,  # noqa: E731
                                 beta=beta if with_beta else None)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20 * 1e3, dw, dg, db


t_old, dw0, dg0, db0 = run(False)
t_new, dw1, dg1, db1 = run(True)
rel = lambda a, b: float((a - b).norm() / b.norm())  # noqa: E731
print("gate read %.1f us | gate worked out again (FC_GATE=%s WGS=%s) %.1f us (reduce + fold + final) | rel diff dw %.2e "
      "dgamma %.2e dbeta %.2e" % (t_old, os.environ.get("ATHENA_FC_GATE", "3"), os.environ.get("ATHENA_FC_WGS", "768"),
                                  t_new, rel(dw1, dw0), rel(dg1, dg0), rel(db1, db0)))
