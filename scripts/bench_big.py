"""A/B timings of the 256x256 direct-to-LDS GEMM core against the 128x128 core (GPU scratch driver):
the three conv2 products at the AISHELL size and the encoder's Dense weight gradients."""
import os
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops
from athena_amd.layers import fused  # noqa: F401

dev = torch.device("cuda", 0)
cd = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for n in (4096, 8192):
    a = torch.randn(n, n, device=dev).to(cd)
    b = torch.randn(n, n, device=dev).to(cd)
    c = torch.empty(n, n, device=dev, dtype=cd)
    for mode in ("0", "1"):
        os.environ["ATHENA_GEMM_BIG"] = mode
        t = timeit(lambda: ops.gemm(a, b, c, n, n, n, trans_a=False, trans_b=True, ldb=n, ldc=n), n=10)
        print("BIG=%s plain %d^3: %.1f us (%.0f TF/s)" % (mode, n, t, 2.0 * n ** 3 / t / 1e6))
del a, b, c
B, H, W, C = 32, 500, 40, 256
Ho, Wo = 250, 20
x = torch.randn(B, H, W, C, device=dev).to(cd)
w_t = (torch.randn(C, 3, 3, C, device=dev) * 0.05).to(cd)
w_hwio = (torch.randn(3, 3, C, C, device=dev) * 0.05).to(cd)
dy = torch.randn(B, Ho, Wo, C, device=dev).to(cd)
out = torch.empty(B, Ho, Wo, C, device=dev, dtype=cd)
din = torch.empty(B, H, W, C, device=dev, dtype=cd)
dw = torch.empty(3, 3, C, C, device=dev)
stats = torch.empty(2 * C, dtype=torch.float64, device=dev)
fl = 2.0 * 9 * C * C * B * Ho * Wo
for mode in ("0", "1"):
    os.environ["ATHENA_GEMM_BIG"] = mode
    t1 = timeit(lambda: ops.conv3x3s2_fwd(x, w_t, 1, out, B, H, W, C, C, stats))
    t2 = timeit(lambda: ops.conv3x3s2_dgrad(dy, w_hwio, din, B, H, W, C, C))
    t3 = timeit(lambda: ops.conv3x3s2_wgrad(x, dy, dw, B, H, W, C, C))
    print("BIG=%s conv2 fwd %.1f us (%.0f TF/s)  dgrad %.1f us (%.0f)  wgrad %.1f us (%.0f)" %
          (mode, t1, fl / t1 / 1e6, t2, fl / t2 / 1e6, t3, fl / t3 / 1e6))

# encoder weight gradients: 12 layers x (q, k, v, o: 256x256; ffn0 256x1024; ffn1 1024x256), 8000 rows
rows, d, dff, L = 8000, 256, 1024, 12
probs_tab, probs_grp = [], []
for layer in range(L):
    xs = torch.randn(rows, d, device=dev).to(cd)
    h = torch.randn(rows, dff, device=dev).to(cd)
    dqkv = torch.randn(rows, 3 * d, device=dev).to(cd)
    do = torch.randn(rows, d, device=dev).to(cd)
    dh = torch.randn(rows, dff, device=dev).to(cd)
    df = torch.randn(rows, d, device=dev).to(cd)
    layer_t, layer_g = [], []
    for (a, b, M, N, lda, ldb) in [(xs, dqkv[:, :d], d, d, d, 3 * d), (xs, dqkv[:, d:2 * d], d, d, d, 3 * d),
                                   (xs, dqkv[:, 2 * d:], d, d, d, 3 * d), (xs, do, d, d, d, d),
                                   (xs, dh, d, dff, d, dff), (h, df, dff, d, dff, d)]:
        G = torch.empty(M, N, device=dev)
        gb = torch.empty(N, device=dev)
        layer_t.append((a, b, rows, M, N, lda, ldb, G, gb))
        layer_g.append((a, b, G, M, N, rows, dict(trans_a=True, trans_b=False, lda=lda, ldb=ldb, ldc=N, split_k=0,
                                                  b_colsum=gb)))
    probs_tab += layer_t
    probs_grp.append(layer_g)
fl = 2.0 * rows * L * (4 * d * d + 2 * d * dff)
t_tab = timeit(lambda: ops.gemm_wgrad_table(probs_tab))
t_grp = timeit(lambda: [ops.gemm_grouped(g) for g in probs_grp])
print("dense weight gradients of 12 layers: one table launch %.1f us (%.0f TF/s), 12 grouped launches %.1f us (%.0f)" %
      (t_tab, fl / t_tab / 1e6, t_grp, fl / t_grp / 1e6))
