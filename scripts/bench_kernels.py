"""Micro-benchmarks of single entry points on the GPU (HIP events, median of repeated launches)."""
import math
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)


def timeit(fn, reps=20, warm=3):
    """Average device time of one call: ``reps`` launches captured in a HIP graph and replayed,
    so host launch overhead is excluded."""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2] * 1e-3 / reps


def bench_gemm():
    print("== gemm bf16 (M, N, K, ta, tb, out, act) ==")
    cases = [
        (8000, 1024, 256, False, True, torch.bfloat16, ops.ACT_NONE, "ffn1 plain"),
        (8000, 1024, 256, False, True, torch.bfloat16, ops.ACT_GELU, "ffn1 gelu+pre+drop"),
        (8000, 256, 1024, False, True, torch.bfloat16, ops.ACT_NONE, "ffn2"),
        (8000, 256, 256, False, True, torch.bfloat16, ops.ACT_NONE, "proj"),
        (8000, 4233, 256, False, True, torch.float32, ops.ACT_NONE, "vocab fwd"),
        (8000, 256, 4240, False, True, torch.bfloat16, ops.ACT_NONE, "vocab dx"),
        (8000, 256, 5120, False, True, torch.bfloat16, ops.ACT_RELU6, "x-dense"),
        (1056, 256, 256, False, True, torch.bfloat16, ops.ACT_NONE, "dec proj"),
        (1056, 4233, 256, False, True, torch.float32, ops.ACT_NONE, "dec vocab"),
        (256, 1024, 8000, True, False, torch.float32, ops.ACT_NONE, "dW ffn1"),
        (256, 256, 8000, True, False, torch.float32, ops.ACT_NONE, "dW proj"),
        (256, 4233, 8000, True, False, torch.float32, ops.ACT_NONE, "dW vocab"),
        (5120, 256, 8000, True, False, torch.float32, ops.ACT_NONE, "dW x-dense"),
        (256, 256, 1056, True, False, torch.float32, ops.ACT_NONE, "dW dec proj"),
    ]
    rng = torch.tensor([1, 2], dtype=torch.int64, device=dev)
    for M, N, K, ta, tb, od, act, name in cases:
        a = torch.randn((K, M) if ta else (M, K), device=dev).to(torch.bfloat16)
        ldb = (K if tb else ops.roundup(N, 8))
        b = torch.randn((N, ldb) if tb else (K, ldb), device=dev).to(torch.bfloat16)
        ldc = ops.roundup(N, 8) if (od == torch.float32 and not ta) else N
        c = torch.empty((M, ldc), dtype=od, device=dev)
        bias = torch.randn(N, device=dev) if not ta else None
        pre = torch.empty_like(c) if act == ops.ACT_GELU else None
        kw = dict(trans_a=ta, trans_b=tb, ldb=ldb, ldc=ldc, bias=bias, act=act, c_pre=pre)
        if act == ops.ACT_GELU:
            kw.update(drop_p=0.1, rng_state=rng, rng_site=3)
        if ta:
            kw.update(split_k=0)
        t = timeit(lambda: ops.gemm(a, b, c, M, N, K, **kw))
        print("%-20s M=%5d N=%5d K=%5d  %8.1f us  %7.1f TFLOP/s" % (name, M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12))


def bench_conv():
    print("== conv2 3x3 s2 (B=32, 500x40x256 -> 250x20x256) bf16 ==")
    B, H, W, C = 32, 500, 40, 256
    x = torch.randn(B, H, W, C, device=dev).to(torch.bfloat16)
    w = (torch.randn(3, 3, C, C, device=dev) * 0.05).to(torch.bfloat16)
    wt = w.permute(3, 0, 1, 2).contiguous()
    Ho, Wo = 250, 20
    out = torch.empty(B, Ho, Wo, C, device=dev, dtype=torch.bfloat16)
    st = torch.empty(2 * C, dtype=torch.float64, device=dev)
    fl = 2.0 * 9 * C * C * B * Ho * Wo
    t = timeit(lambda: ops.conv3x3s2_fwd(x, wt, 1, out, B, H, W, C, C, st), reps=5)
    print("fwd   %8.1f us %7.1f TFLOP/s" % (t * 1e6, fl / t / 1e12))
    t = timeit(lambda: ops.conv3x3s2_fwd(x, wt, 1, out, B, H, W, C, C, None), reps=5)
    print("fwd (no column statistics) %8.1f us %7.1f TFLOP/s" % (t * 1e6, fl / t / 1e12))
    dy = torch.randn_like(out)
    din = torch.empty_like(x)
    t = timeit(lambda: ops.conv3x3s2_dgrad(dy, w, din, B, H, W, C, C), reps=5)
    print("dgrad %8.1f us %7.1f TFLOP/s" % (t * 1e6, fl / t / 1e12))
    dw = torch.empty(3, 3, C, C, device=dev)
    t = timeit(lambda: ops.conv3x3s2_wgrad(x, dy, dw, B, H, W, C, C), reps=3, warm=1)
    print("wgrad %8.1f us %7.1f TFLOP/s" % (t * 1e6, fl / t / 1e12))


def bench_fc():
    print("== first layer conv(1->256)+BN+relu6 fused (B=32, 1000x80 -> 500x40x256) bf16 ==")
    B, H, W, C = 32, 1000, 80, 256
    Ho, Wo = 500, 40
    x = torch.randn(B, H, W, device=dev)
    w = torch.randn(9, C, device=dev) * 0.3
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mom = torch.zeros(ops.conv1_moments_len(), dtype=torch.float64, device=dev)
    st = torch.zeros(2 * C, dtype=torch.float64, device=dev)
    mr = torch.empty(2 * C, device=dev)
    out = torch.empty(B * Ho * Wo, C, device=dev, dtype=torch.bfloat16)
    t = timeit(lambda: ops.conv1_bn_stats(x, w, B, H, W, C, mom, st), reps=5)
    print("moments + column statistics %8.1f us" % (t * 1e6))
    ops.bn_finalize(st, B * Ho * Wo, C, 1e-3, 0.99, mr, None, None)
    t = timeit(lambda: ops.conv1_bn_relu6_fwd(x, w, mr, None, gamma, beta, 1e-3, False, B, H, W, C, out), reps=5)
    by = out.numel() * 2
    print("forward  %8.1f us (%5.0f GB/s written)" % (t * 1e6, by / t / 1e9))
    g = torch.randn_like(out)
    dw, dg, db = torch.empty(9, C, device=dev), torch.empty(C, device=dev), torch.empty(C, device=dev)
    t = timeit(lambda: ops.conv1_bn_bwd(x, g, w, mom, mr, gamma, B, H, W, C, dw, dg, db, relu6_out=out), reps=5)
    print("backward %8.1f us (%5.0f GB/s read)" % (t * 1e6, 2 * by / t / 1e9))


def bench_attn():
    print("== attention B=32 h=4 T=250 dk=64 bf16 ==")
    B, h, T, dk = 32, 4, 250, 64
    d = h * dk
    q, k, v, do = (torch.randn(B, T, h, dk, device=dev).to(torch.bfloat16) for _ in range(4))
    km = torch.zeros(B, T, dtype=torch.uint8, device=dev)
    out = torch.empty_like(q)
    lse = torch.empty(B, h, T, device=dev)
    fl = 4.0 * B * h * T * T * dk
    t = timeit(lambda: ops.mha_fwd(q, k, v, d, d, d, km, False, B, h, T, T, dk, 0.125, out, d, lse))
    print("fwd  %8.1f us %7.1f TFLOP/s" % (t * 1e6, fl / t / 1e12))
    dq, dk_, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    delta = torch.empty_like(lse)
    t = timeit(lambda: ops.mha_bwd(q, k, v, d, d, d, km, False, B, h, T, T, dk, 0.125, out, do, d, lse, delta, dq,
                                   dk_, dv, d, d, d))
    print("bwd  %8.1f us %7.1f TFLOP/s" % (t * 1e6, 2.5 * fl / t / 1e12))


def bench_ctc():
    # the benchmark configuration, the LibriSpeech-long and the CTC-heavy configurations
    for B, T, V, U in [(32, 250, 4233, 32), (8, 750, 5001, 150), (64, 250, 8001, 200)]:
        print("== CTC B=%d T'=%d V=%d U=%d fp32 ==" % (B, T, V, U))
        ldl = ops.roundup(V, 8)
        logits = torch.randn(B, T, ldl, device=dev)
        labels = torch.randint(1, V - 1, (B, U), device=dev, dtype=torch.int32)
        ll = torch.full((B,), U, dtype=torch.int32, device=dev)
        tl = torch.full((B,), T, dtype=torch.int32, device=dev)
        ws = torch.empty(ops.ctc_workspace_bytes(B, T, U), dtype=torch.uint8, device=dev)
        per = torch.empty(B, device=dev)
        grad = torch.empty_like(logits)
        t1 = timeit(lambda: ops.ctc_loss_fwd(logits, ldl, labels, ll, tl, B, T, V, U, V - 1, per, ws))
        t2 = timeit(lambda: ops.ctc_loss_bwd(logits, ldl, labels, ll, tl, B, T, V, U, V - 1, None, 1.0 / B, grad, ws))
        g16 = torch.empty(B, T, ldl, device=dev, dtype=torch.bfloat16)
        t3 = timeit(lambda: ops.ctc_loss_fwd_bwd(logits, ldl, labels, ll, tl, B, T, V, U, V - 1, per, ws, 1.0 / B, g16))
        t4 = timeit(lambda: ops.ctc_loss_fwd_bwd(logits, ldl, labels, ll, tl, B, T, V, U, V - 1, per, ws, 1.0 / B, grad))
        by = B * T * V * 4
        print("one pass, bf16 gradient %8.1f us (%6.0f GB/s of the %d MB two passes would move)   fp32 gradient %8.1f us" %
              (t3 * 1e6, 2.5 * by / t3 / 1e9, int(2.5 * by) >> 20, t4 * 1e6))
        print("fwd %8.1f us (%6.0f GB/s read)   bwd %8.1f us (%6.0f GB/s r+w)   total %6.0f GB/s of %d MB" %
              (t1 * 1e6, by / t1 / 1e9, t2 * 1e6, 2 * by / t2 / 1e9, 3 * by / (t1 + t2) / 1e9, 3 * by >> 20))


if __name__ == "__main__":
    which = sys.argv[1:] or ["gemm", "conv", "attn", "ctc"]
    for w in which:
        globals()["bench_" + w]()
