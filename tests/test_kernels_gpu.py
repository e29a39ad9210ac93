"""Parity of each HIP entry point (called through the C ABI) against the CPU oracle on seeded
inputs.  fp32 kernels must agree to ~1e-5 relative; bf16 storage to ~1e-2 (8 bits of mantissa)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ctc as octc
from oracle import model as om
from tests.util import TOL, assert_close, dev, rel_err, rel_l2, rounded, to_cd

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from athena_amd import ops as o
    return o


DTYPES = [torch.float32, torch.bfloat16]


# ------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("cd", DTYPES)
@pytest.mark.parametrize("ta,tb", [(False, True), (False, False), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(200, 72, 64), (136, 264, 96), (1000, 520, 40), (64, 64, 512)])
def test_gemm_layouts(ops, cd, ta, tb, M, N, K):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K + ta * 2 + tb)
    A = torch.randn((K, M) if ta else (M, K), generator=g, dtype=torch.float64)
    Bm = torch.randn((N, K) if tb else (K, N), generator=g, dtype=torch.float64)
    Ar, Br = rounded(A, cd), rounded(Bm, cd)
    ref = (Ar.t() if ta else Ar) @ (Br.t() if tb else Br)
    C = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev())
    ops.gemm(to_cd(A, cd), to_cd(Bm, cd), C, M, N, K, trans_a=ta, trans_b=tb)
    assert_close(C, ref, 2e-5 if cd == torch.float32 else 1e-5, "gemm")   # products of rounded inputs are exact in fp32 acc


@pytest.mark.parametrize("M,N", [(4420, 3700), (4500, 3590), (1030, 27000)])
def test_gemm_large_tiles_with_a_halved_last_round(ops, M, N):
    """Plain bf16 products big enough for the 256 x 256 core whose tile count leaves a last round of at
    most half a chip: those tiles run as 128-row halves (a half that starts beyond the last row
    included).  Against the fp64 product of the rounded operands, with a bias, fp32 and bf16 outputs."""
    K = 256
    g = torch.Generator().manual_seed(M + N)
    A = torch.randn(M, K, generator=g, dtype=torch.float64)
    Bm = torch.randn(N, K, generator=g, dtype=torch.float64)
    bias = torch.randn(N, generator=g, dtype=torch.float64)
    ref = rounded(A, torch.bfloat16) @ rounded(Bm, torch.bfloat16).t() + bias
    a, b = to_cd(A, torch.bfloat16), to_cd(Bm, torch.bfloat16)
    C = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev())
    ops.gemm(a, b, C, M, N, K, trans_b=True, bias=bias.float().to(dev()))
    assert_close(C, ref, 1e-5, "fp32 output")
    C16 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev())
    ops.gemm(a, b, C16, M, N, K, trans_b=True, bias=bias.float().to(dev()))
    assert_close(C16, ref, 5e-3, "bf16 output")


@pytest.mark.parametrize("cd", DTYPES)
def test_gemm_epilogues(ops, cd):
    M, N, K = 150, 88, 64
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g, dtype=torch.float64)
    W = torch.randn(K, N, generator=g, dtype=torch.float64) * 0.3
    bias = torch.randn(N, generator=g, dtype=torch.float64)
    Ar, Wr = rounded(A, cd), rounded(W, cd)
    pre_ref = Ar @ Wr + bias
    tol = TOL[cd]
    # bias + gelu with pre-activation copy
    C = torch.empty(M, N, dtype=cd, device=dev())
    P = torch.empty(M, N, dtype=cd, device=dev())
    ops.gemm(to_cd(A, cd), to_cd(W, cd), C, M, N, K, bias=bias.float().to(dev()), act=ops.ACT_GELU, c_pre=P)
    assert_close(P, pre_ref, tol, "pre-activation")
    assert_close(C, om.gelu_tanh(pre_ref), tol, "gelu")
    # relu6
    ops.gemm(to_cd(A, cd), to_cd(W, cd), C, M, N, K, bias=bias.float().to(dev()), act=ops.ACT_RELU6)
    assert_close(C, om.relu6(pre_ref), tol, "relu6")
    # backward epilogue: multiply by gelu'(aux)
    H = torch.randn(M, N, generator=g, dtype=torch.float64)
    Hr = rounded(H, cd).requires_grad_(True)
    om.gelu_tanh(Hr).sum().backward()
    ops.gemm(to_cd(A, cd), to_cd(W, cd), C, M, N, K, act=ops.ACT_MUL_DGELU, aux=to_cd(H, cd), ldaux=N)
    assert_close(C, (Ar @ Wr) * Hr.grad, tol, "mul_dgelu")
    # relu6 gate from the saved output
    Y = om.relu6(torch.randn(M, N, generator=g, dtype=torch.float64) * 4)
    Yr = rounded(Y, cd)
    ops.gemm(to_cd(A, cd), to_cd(W, cd), C, M, N, K, act=ops.ACT_MUL_DRELU6, aux=to_cd(Y, cd), ldaux=N)
    assert_close(C, (Ar @ Wr) * ((Yr > 0) & (Yr < 6)).double(), tol, "mul_drelu6")
    # accumulate
    C0 = torch.randn(M, N, generator=g, dtype=torch.float64)
    C = to_cd(C0, torch.float32)
    ops.gemm(to_cd(A, cd), to_cd(W, cd), C, M, N, K, accumulate=True)
    assert_close(C, Ar @ Wr + C0.float().double(), 2e-5, "accumulate")
    # automatic split-K (the weight-gradient shape: long contraction, small output)
    X = torch.randn(3000, 72, generator=g, dtype=torch.float64)
    dY = torch.randn(3000, 48, generator=g, dtype=torch.float64)
    G = torch.full((72, 48), float("nan"), dtype=torch.float32, device=dev())
    ops.gemm(to_cd(X, cd), to_cd(dY, cd), G, 72, 48, 3000, trans_a=True, trans_b=False, split_k=0)
    assert_close(G, rounded(X, cd).t() @ rounded(dY, cd), 2e-5, "split-k")


@pytest.mark.parametrize("cd", DTYPES)
@pytest.mark.parametrize("rows,K,N,ldx,ldy", [(3000, 72, 48, 72, 48), (8000, 256, 256, 256, 256),
                                              (1056, 256, 4233, 256, 4240), (500, 1024, 64, 1024, 64),
                                              (4000, 136, 100, 136, 104), (64, 5120, 256, 5120, 256)])
def test_gemm_weight_gradient(ops, cd, rows, K, N, ldx, ldy):
    """x^T.dy with automatic split-K through the workspace, bias gradient from the same launch."""
    g = torch.Generator().manual_seed(rows + K + N)
    X = torch.randn(rows, ldx, generator=g, dtype=torch.float64)
    dY = torch.randn(rows, ldy, generator=g, dtype=torch.float64)
    Xr, dYr = rounded(X, cd)[:, :K], rounded(dY, cd)[:, :N]
    G = torch.full((K, N), float("nan"), dtype=torch.float32, device=dev())
    gb = torch.full((N,), float("nan"), dtype=torch.float32, device=dev())
    x_d, dy_d = to_cd(X, cd), to_cd(dY, cd)
    ops.gemm(x_d, dy_d, G, K, N, rows, trans_a=True, trans_b=False, lda=ldx, ldb=ldy, split_k=0, b_colsum=gb)
    assert_close(G, Xr.t() @ dYr, 2e-5, "weight gradient")
    assert_close(gb, dYr.sum(0), 2e-5, "bias gradient")
    G2, gb2 = torch.empty_like(G), torch.empty_like(gb)
    ops.gemm(x_d, dy_d, G2, K, N, rows, trans_a=True, trans_b=False, lda=ldx, ldb=ldy, split_k=0, b_colsum=gb2)
    if cd == torch.bfloat16:      # fixed-order reduction: bit-reproducible
        assert torch.equal(G, G2) and torch.equal(gb, gb2)


@pytest.mark.parametrize("cd", DTYPES)
def test_gemm_grouped(ops, cd):
    """Grouped launches give what the single launches give: weight gradients (split K, bias sums)
    of different shapes, and projections with bias."""
    g = torch.Generator().manual_seed(77)
    shapes = [(2000, 256, 256), (2000, 256, 1024), (2000, 1024, 256), (500, 256, 256), (2000, 72, 40), (330, 64, 4233)]
    single, grouped, probs, keep = [], [], [], []
    for rows, K, N in shapes:
        ldy = (N + 7) // 8 * 8
        x = to_cd(torch.randn(rows, K, generator=g, dtype=torch.float64), cd)
        dy = to_cd(torch.randn(rows, ldy, generator=g, dtype=torch.float64), cd)
        G1, b1 = torch.empty(K, N, device=dev()), torch.empty(N, device=dev())
        G2 = torch.full((K, N), float("nan"), device=dev())
        b2 = torch.full((N,), float("nan"), device=dev())
        kw = dict(trans_a=True, trans_b=False, lda=K, ldb=ldy, ldc=N, split_k=0)
        ops.gemm(x, dy, G1, K, N, rows, b_colsum=b1, **kw)
        probs.append((x, dy, G2, K, N, rows, dict(b_colsum=b2, **kw)))
        single.append((G1, b1))
        grouped.append((G2, b2))
    ops.gemm_grouped(probs)
    for (G1, b1), (G2, b2) in zip(single, grouped):
        assert_close(G2, G1.double().cpu(), 1e-6, "grouped weight gradient")
        assert_close(b2, b1.double().cpu(), 1e-6, "grouped bias gradient")
    # projections: same x, three weights, bias
    M, K, N = 1000, 256, 256
    x = to_cd(torch.randn(M, K, generator=g, dtype=torch.float64), cd)
    probs, refs = [], []
    for i in range(3):
        w = to_cd(torch.randn(N, K, generator=g, dtype=torch.float64) * 0.1, cd)
        b = torch.randn(N, generator=g).to(dev())
        o1 = torch.empty(M, N, dtype=cd, device=dev())
        o2 = torch.full((M, N), float("nan"), dtype=cd, device=dev())
        ops.gemm(x, w, o1, M, N, K, trans_b=True, bias=b)
        probs.append((x, w, o2, M, N, K, dict(trans_b=True, bias=b)))
        refs.append((o1, o2))
    ops.gemm_grouped(probs)
    for o1, o2 in refs:
        assert torch.equal(o1, o2)


@pytest.mark.parametrize("cd", DTYPES)
def test_gemm_dropout_consistency(ops, cd):
    """Same site + state gives the same mask in the forward and the backward epilogue."""
    M, N, K = 96, 64, 32
    g = torch.Generator().manual_seed(9)
    A = to_cd(torch.randn(M, K, generator=g, dtype=torch.float64), cd)
    W = to_cd(torch.randn(K, N, generator=g, dtype=torch.float64), cd)
    rng = torch.tensor([1234, 7], dtype=torch.int64, device=dev())
    C0 = torch.empty(M, N, dtype=torch.float32, device=dev())
    C1 = torch.empty(M, N, dtype=torch.float32, device=dev())
    C2 = torch.empty(M, N, dtype=torch.float32, device=dev())
    ops.gemm(A, W, C0, M, N, K)
    ops.gemm(A, W, C1, M, N, K, drop_p=0.25, rng_state=rng, rng_site=3)
    ops.gemm(A, W, C2, M, N, K, drop_p=0.25, rng_state=rng, rng_site=3)
    assert torch.equal(C1, C2)
    kept = C1 != 0
    frac = float(kept.float().mean())
    assert 0.70 < frac < 0.80, frac
    assert_close(C1[kept], C0[kept] / 0.75, 1e-6, "dropout scale")
    ops.gemm(A, W, C2, M, N, K, drop_p=0.25, rng_state=rng, rng_site=4)
    assert not torch.equal(C1, C2)


@pytest.mark.parametrize("cd", DTYPES)
def test_colsum(ops, cd):
    g = torch.Generator().manual_seed(2)
    for M, N, ld in [(300, 40, 40), (1000, 4233, 4240), (77, 8, 16)]:
        X = torch.randn(M, ld, generator=g, dtype=torch.float64)
        out = torch.empty(N, dtype=torch.float32, device=dev())
        ops.colsum(to_cd(X, cd), M, N, ld, out)
        assert_close(out, rounded(X, cd)[:, :N].sum(0), 1e-5, "colsum")


# ------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize("cd", DTYPES)
@pytest.mark.parametrize("rows,d", [(37, 32), (260, 256), (50, 512), (9, 1000)])
def test_add_layernorm(ops, cd, rows, d):
    g = torch.Generator().manual_seed(rows + d)
    a = torch.randn(rows, d, generator=g, dtype=torch.float64)
    b = torch.randn(rows, d, generator=g, dtype=torch.float64)
    gamma = torch.rand(d, generator=g, dtype=torch.float64) + 0.5
    beta = torch.randn(d, generator=g, dtype=torch.float64)
    dy = torch.randn(rows, d, generator=g, dtype=torch.float64)
    ar, br = rounded(a, cd), rounded(b, cd)
    zr = rounded(ar + br, cd).requires_grad_(True)       # the kernel stores z in the compute dtype
    gm, bt = gamma.float().double().requires_grad_(True), beta.float().double().requires_grad_(True)
    y_ref = om.layer_norm(zr, gm, bt, 1e-8)
    (y_ref * rounded(dy, cd)).sum().backward()
    gam_d, bet_d = gamma.float().to(dev()), beta.float().to(dev())
    z, y, mean, rstd = ops.add_layernorm_fwd(to_cd(a, cd), to_cd(b, cd), gam_d, bet_d, 1e-8, 0.0, None, 0)
    tol = TOL[cd]
    assert_close(z, zr.detach(), 1e-6 if cd == torch.float32 else 1e-2, "z")
    assert_close(y, y_ref.detach(), tol, "y")
    dgamma = torch.empty(d, dtype=torch.float32, device=dev())
    dbeta = torch.empty(d, dtype=torch.float32, device=dev())
    da, db = ops.add_layernorm_bwd(to_cd(dy, cd), z, mean, rstd, gam_d, 0.0, None, 0, dgamma, dbeta)
    assert da is db
    assert_close(da, zr.grad, tol * 2, "dz")
    assert_close(dgamma, gm.grad, tol * 2, "dgamma")
    assert_close(dbeta, bt.grad, tol, "dbeta")


def test_add_layernorm_dropout(ops):
    rows, d = 64, 128
    g = torch.Generator().manual_seed(3)
    a = torch.randn(rows, d, generator=g).to(dev())
    b = torch.randn(rows, d, generator=g).to(dev())
    gam, bet = torch.ones(d, device=dev()), torch.zeros(d, device=dev())
    rng = torch.tensor([99, 1], dtype=torch.int64, device=dev())
    z, y, mean, rstd = ops.add_layernorm_fwd(a, b, gam, bet, 1e-8, 0.5, rng, 11)
    m = (z - a) / b                      # 0 or 2
    assert set(torch.unique(torch.round(m)).tolist()) <= {0.0, 2.0}
    dg, dbt = torch.empty(d, device=dev()), torch.empty(d, device=dev())
    dy = torch.randn(rows, d, generator=g).to(dev())
    da, db = ops.add_layernorm_bwd(dy, z, mean, rstd, gam, 0.5, rng, 11, dg, dbt)
    assert_close(db, da * torch.round(m), 1e-6, "db = dz * mask")


@pytest.mark.parametrize("p", [0.1, 0.5])
def test_dropout_decisions_are_fair_and_unrelated(ops, p):
    """The dropout draw hands out four decisions per hash (two from the hash, two from one more mixing
    round on top of it): over 4 M elements the keep rate is 1 - p, the four positions of a group keep
    equally often, decisions inside a group and across neighbouring groups are uncorrelated, and two
    sites or two steps give unrelated masks."""
    rows, d = 4096, 1024
    a = torch.zeros(rows, d, device=dev())
    b = torch.ones(rows, d, device=dev())
    gam, bet = torch.ones(d, device=dev()), torch.zeros(d, device=dev())

    def mask(seed, step, site):
        rng = torch.tensor([seed, step], dtype=torch.int64, device=dev())
        z, _, _, _ = ops.add_layernorm_fwd(a, b, gam, bet, 1e-8, p, rng, site)       # z = dropout(1)
        return (z != 0).double().cpu()

    m = mask(99, 1, 11)
    n = m.numel()
    sd = (p * (1 - p)) ** 0.5
    assert abs(float(m.mean()) - (1 - p)) < 5 * sd / n ** 0.5 + 2e-5             # (+ the 16-bit threshold's rounding)
    quads = m.view(-1, 4)
    for k in range(4):
        assert abs(float(quads[:, k].mean()) - (1 - p)) < 6 * sd / (n / 4) ** 0.5 + 2e-5, k
    c = quads - quads.mean(0)
    cov = (c.t() @ c) / quads.shape[0] / (p * (1 - p))
    off = cov - torch.diag(torch.diag(cov))
    assert float(off.abs().max()) < 6 / (n / 4) ** 0.5, off
    flat = m.view(-1) - m.mean()
    for lag in (4, 5, 8, d, d + 1):
        r = float((flat[:-lag] * flat[lag:]).mean()) / (p * (1 - p))
        assert abs(r) < 6 / n ** 0.5, (lag, r)
    for other in (mask(99, 2, 11), mask(99, 1, 12), mask(100, 1, 11)):
        r = float(((other - other.mean()) * flat.view_as(other)).mean()) / (p * (1 - p))
        assert abs(r) < 6 / n ** 0.5, r


# ------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, key_mask, causal, scale):
    B, Tq, h, dk = q.shape
    Tk = k.shape[1]
    qh, kh, vh = q.permute(0, 2, 1, 3), k.permute(0, 2, 1, 3), v.permute(0, 2, 1, 3)
    logits = qh @ kh.transpose(-1, -2) * scale
    mask = torch.zeros(B, 1, Tq, Tk, dtype=q.dtype)
    if key_mask is not None:
        mask = torch.maximum(mask, key_mask.to(q.dtype)[:, None, None, :])
    if causal:
        # True / 1: the look-ahead mask; c > 1: the band of a unidirectional layer with look_ahead = c - 1
        mask = torch.maximum(mask, 1 - torch.tril(torch.ones(Tq, Tk, dtype=q.dtype), diagonal=int(causal) - 1))
    logits = logits + mask * -1e9
    w = torch.softmax(logits, -1)
    return (w @ vh).permute(0, 2, 1, 3)


@pytest.mark.parametrize("cd", DTYPES)
@pytest.mark.parametrize("B,h,Tq,Tk,dk,causal", [(2, 4, 37, 37, 8, False), (3, 2, 19, 19, 32, True),
                                                (2, 4, 33, 70, 64, False), (1, 8, 130, 130, 32, False),
                                                (2, 2, 5, 5, 16, True), (2, 2, 250, 250, 64, False),
                                                (1, 2, 33, 250, 64, False), (1, 1, 70, 70, 128, True),
                                                (2, 4, 200, 200, 64, True), (1, 2, 64, 128, 32, False),
                                                # LibriSpeech-long: 750 frames after the front end, 8 heads of 64;
                                                # decoder cross-attention of 151 label positions over them
                                                (2, 8, 750, 750, 64, False), (2, 8, 151, 750, 64, False),
                                                (1, 8, 151, 151, 64, True),
                                                # unidirectional encoder layers: look_ahead 3 and 2
                                                (2, 4, 100, 100, 64, 4), (2, 2, 37, 37, 8, 3)])
def test_mha(ops, cd, B, h, Tq, Tk, dk, causal):
    g = torch.Generator().manual_seed(B * 100 + Tq + dk)
    d = h * dk
    q = torch.randn(B, Tq, h, dk, generator=g, dtype=torch.float64)
    k = torch.randn(B, Tk, h, dk, generator=g, dtype=torch.float64)
    v = torch.randn(B, Tk, h, dk, generator=g, dtype=torch.float64)
    do = torch.randn(B, Tq, h, dk, generator=g, dtype=torch.float64)
    key_mask = torch.zeros(B, Tk, dtype=torch.uint8)
    for b in range(B):
        n = int(torch.randint(1, Tk + 1, (1,), generator=g))
        key_mask[b, n:] = 1
    if causal:
        key_mask[:, 0] = 0
        if Tk > 3:
            key_mask[0, 2] = 1      # a masked key in the middle (label id 0 inside a sequence)
    scale = 1.0 / math.sqrt(dk)
    qr, kr, vr = (rounded(t, cd).requires_grad_(True) for t in (q, k, v))
    ref = _attn_ref(qr, kr, vr, key_mask, causal, scale)
    (ref * rounded(do, cd)).sum().backward()
    qd, kd, vd, dod = (to_cd(t, cd) for t in (q, k, v, do))
    out = torch.empty(B, Tq, h, dk, dtype=cd, device=dev())
    lse = torch.empty(B, h, Tq, dtype=torch.float32, device=dev())
    km = key_mask.to(dev())
    ops.mha_fwd(qd, kd, vd, d, d, d, km, causal, B, h, Tq, Tk, dk, scale, out, d, lse)
    tol = 1e-4 if cd == torch.float32 else 2e-2
    assert_close(out, ref.detach(), tol, "attention out")
    dq, dk_, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    delta = torch.empty_like(lse)
    # the backward is fed the oracle's (rounded) forward output so the two passes are tested apart
    out_in = to_cd(ref.detach(), cd)
    ops.mha_bwd(qd, kd, vd, d, d, d, km, causal, B, h, Tq, Tk, dk, scale, out_in, dod, d, lse, delta, dq, dk_, dv,
                d, d, d)
    assert_close(dq, qr.grad, tol * 2, "dq")
    assert_close(dk_, kr.grad, tol * 2, "dk")
    assert_close(dv, vr.grad, tol * 2, "dv")


def test_mha_fully_masked_row_is_uniform(ops):
    """mask * -1e9 in fp32 swallows the scores: a row whose keys are all masked averages V."""
    B, h, T, dk = 1, 1, 6, 8
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(B, T, h, dk, generator=g).to(dev()) for _ in range(3))
    km = torch.ones(B, T, dtype=torch.uint8, device=dev())
    out = torch.empty_like(q)
    lse = torch.empty(B, h, T, device=dev())
    ops.mha_fwd(q, k, v, dk, dk, dk, km, False, B, h, T, T, dk, 1.0, out, dk, lse)
    assert_close(out[0, 0, 0], v[0, :, 0].mean(0), 1e-5, "uniform")


def test_mha_fully_masked_row_is_uniform_bf16(ops):
    """Same property on the matrix-core kernels (bf16, head size 64, more keys than one block)."""
    B, h, T, dk = 2, 2, 100, 64
    g = torch.Generator().manual_seed(1)
    q, k, v = (torch.randn(B, T, h, dk, generator=g).to(dev()).to(torch.bfloat16) for _ in range(3))
    km = torch.ones(B, T, dtype=torch.uint8, device=dev())
    km[1, :40] = 0
    out = torch.empty_like(q)
    lse = torch.empty(B, h, T, device=dev())
    ops.mha_fwd(q, k, v, h * dk, h * dk, h * dk, km, False, B, h, T, T, dk, 0.125, out, h * dk, lse)
    assert_close(out[0, 5, 1], v[0, :, 1].float().mean(0), 1e-2, "uniform")
    assert bool(torch.isfinite(out.float()).all())


@pytest.mark.parametrize("cd", DTYPES)
@pytest.mark.parametrize("B,H,W,C", [(2, 13, 10, 32), (3, 40, 16, 64), (1, 7, 5, 8), (2, 50, 21, 256), (2, 61, 30, 128),
                                     (1, 33, 18, 512)])
def test_conv1_bn_relu6_fused(ops, cd, B, H, W, C):
    """One-channel Conv2D -> BatchNorm (batch statistics) -> relu6 without the pre-activation tensor,
    forward and the three parameter gradients, against autograd on the unfused fp64 chain."""
    g = torch.Generator().manual_seed(H * W + C)
    x = torch.randn(B, H, W, 1, generator=g, dtype=torch.float64) * 1.5 + 0.3
    w = (torch.randn(3, 3, 1, C, generator=g, dtype=torch.float64) * 0.8).float().double().requires_grad_(True)
    gamma = (torch.rand(C, generator=g, dtype=torch.float64) + 0.5).float().double().requires_grad_(True)
    beta = (torch.randn(C, generator=g, dtype=torch.float64) * 2 + 2).float().double().requires_grad_(True)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    x = x.float().double()
    pre = om.conv2d_same_s2(x, w)
    mean = pre.mean((0, 1, 2))
    var = pre.var((0, 1, 2), unbiased=False)
    y_ref = om.relu6((pre - mean) / torch.sqrt(var + 1e-3) * gamma + beta)
    dy = torch.randn(B, Ho, Wo, C, generator=g, dtype=torch.float64)
    xd, wd = x.float().to(dev()), w.detach().float().to(dev())
    gd, bd = gamma.detach().float().to(dev()), beta.detach().float().to(dev())
    mom = torch.empty(ops.conv1_moments_len(), dtype=torch.float64, device=dev())
    stats = torch.empty(2 * C, dtype=torch.float64, device=dev())
    ops.conv1_bn_stats(xd, wd, B, H, W, C, mom, stats)
    n = B * Ho * Wo
    assert_close(stats[:C], pre.detach().reshape(-1, C).sum(0), 1e-9, "sum of pre-activations")
    assert_close(stats[C:], (pre.detach() ** 2).reshape(-1, C).sum(0), 1e-9, "sum of squares")
    mr = torch.empty(2 * C, dtype=torch.float32, device=dev())
    ops.bn_finalize(stats, n, C, 1e-3, 0.99, mr, None, None)
    out = torch.full((B, Ho, Wo, C), float("nan"), dtype=cd, device=dev())
    ops.conv1_bn_relu6_fwd(xd, wd, mr, None, gd, bd, 1e-3, False, B, H, W, C, out)
    assert_close(out, y_ref.detach(), 1e-5 if cd == torch.float32 else 8e-3, "fused forward")
    out_train = out.clone()
    # the stored activation carries the exact gate
    gate_ref = ((y_ref > 0) & (y_ref < 6)).detach()
    o64 = out.double().cpu()
    near = ((y_ref.detach() - 6).abs() < 1e-4) | (y_ref.detach().abs() < 1e-4)      # fp32 vs fp64 ties
    assert bool((((o64 > 0) & (o64 < 6)) == gate_ref)[~near].all())
    # backward: the producer hands over the gated gradient
    gated = rounded(dy, cd) * gate_ref.double()
    (y_ref * rounded(dy, cd)).sum().backward()
    dw = torch.full((3, 3, 1, C), float("nan"), dtype=torch.float32, device=dev())
    dg = torch.full((C,), float("nan"), dtype=torch.float32, device=dev())
    db = torch.full((C,), float("nan"), dtype=torch.float32, device=dev())
    ops.conv1_bn_bwd(xd, to_cd(gated, cd), wd, mom, mr, gd, B, H, W, C, dw, dg, db)
    assert_close(db, beta.grad, 2e-5, "dbeta")
    assert_close(dg, gamma.grad, 5e-5, "dgamma")
    assert_close(dw, w.grad, 5e-5, "dw")
    # the same with the gate applied inside: ungated gradient + the stored activation
    dw2, dg2, db2 = torch.empty_like(dw), torch.empty_like(dg), torch.empty_like(db)
    ops.conv1_bn_bwd(xd, to_cd(dy, cd), wd, mom, mr, gd, B, H, W, C, dw2, dg2, db2, relu6_out=out_train)
    assert_close(db2, beta.grad, 2e-5 if cd == torch.float32 else 2e-2, "dbeta (gate inside)")
    assert_close(dw2, w.grad, 5e-5 if cd == torch.float32 else 2e-2, "dw (gate inside)")
    # and with the gate worked out again from the layer's parameters instead of read (bf16 and whole
    # blocks of 64 channels take that path; elsewhere ``beta`` changes nothing): the same gate, so the
    # same sums up to the order of the additions
    dw3, dg3, db3 = torch.empty_like(dw), torch.empty_like(dg), torch.empty_like(db)
    ops.conv1_bn_bwd(xd, to_cd(dy, cd), wd, mom, mr, gd, B, H, W, C, dw3, dg3, db3, relu6_out=out_train, beta=bd)
    assert_close(db3, db2.double(), 1e-5, "dbeta (gate worked out again)")
    assert_close(dg3, dg2.double(), 2e-4, "dgamma (gate worked out again)")
    assert_close(dw3, dw2.double(), 2e-4, "dw (gate worked out again)")
    assert_close(dw3, w.grad, 5e-5 if cd == torch.float32 else 2e-2, "dw (gate worked out again, against autograd)")
    # inference form: moving statistics
    mm = (torch.randn(C, generator=g) * 0.1).to(dev())
    mv = (torch.rand(C, generator=g) + 0.5).to(dev())
    ops.conv1_bn_relu6_fwd(xd, wd, mm, mv, gd, bd, 1e-3, True, B, H, W, C, out)
    y_inf = om.relu6((pre.detach() - mm.double().cpu()) / torch.sqrt(mv.double().cpu() + 1e-3) * gamma.detach() +
                     beta.detach())
    assert_close(out, y_inf, 1e-5 if cd == torch.float32 else 8e-3, "fused inference forward")


# ------------------------------------------------------------------------------------------ conv / BN
@pytest.mark.parametrize("cd", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 12, 10, 16, 24), (1, 13, 7, 8, 16), (3, 20, 9, 32, 136),
                                            (2, 31, 5, 24, 8)])
def test_conv3x3s2(ops, cd, B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(H * W + Cin)
    x = torch.randn(B, H, W, Cin, generator=g, dtype=torch.float64)
    w = torch.randn(3, 3, Cin, Cout, generator=g, dtype=torch.float64) * 0.2
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    dy = torch.randn(B, Ho, Wo, Cout, generator=g, dtype=torch.float64)
    xr, wr = rounded(x, cd).requires_grad_(True), rounded(w, cd).requires_grad_(True)
    ref = om.conv2d_same_s2(xr, wr)
    (ref * rounded(dy, cd)).sum().backward()
    tol = 2e-5 if cd == torch.float32 else 1.5e-2
    xd, dyd = to_cd(x, cd), to_cd(dy, cd)
    w_hwio = to_cd(w, cd)
    w_t = to_cd(w.permute(3, 0, 1, 2), cd)          # [Cout][3][3][Cin]
    stats = torch.empty(2 * Cout, dtype=torch.float64, device=dev())
    for layout, wd in ((0, w_hwio), (1, w_t)):
        out = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=cd, device=dev())
        ops.conv3x3s2_fwd(xd, wd, layout, out, B, H, W, Cin, Cout, stats)
        assert_close(out, ref.detach(), tol, "conv fwd layout %d" % layout)
        o64 = out.double().cpu().reshape(-1, Cout)
        assert_close(stats[:Cout], o64.sum(0), 1e-5, "channel sums")
        assert_close(stats[Cout:], (o64 * o64).sum(0), 1e-5, "channel sums of squares")
    din = torch.full((B, H, W, Cin), float("nan"), dtype=cd, device=dev())
    ops.conv3x3s2_dgrad(dyd, w_hwio, din, B, H, W, Cin, Cout)
    assert_close(din, xr.grad, tol, "conv dgrad")
    # the same with the relu6 gate of the producing layer folded into the epilogue
    act = rounded(om.relu6(torch.randn(B, H, W, Cin, generator=g, dtype=torch.float64) * 4 + 2), cd)
    ops.conv3x3s2_dgrad(dyd, w_hwio, din, B, H, W, Cin, Cout, relu6_out=to_cd(act, cd))
    assert_close(din, xr.grad * ((act > 0) & (act < 6)).double(), tol, "conv dgrad with relu6 gate")
    dw = torch.full((3, 3, Cin, Cout), float("nan"), dtype=torch.float32, device=dev())
    ops.conv3x3s2_wgrad(xd, dyd, dw, B, H, W, Cin, Cout)
    assert_close(dw, wr.grad, tol, "conv wgrad")


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 18, 12, 64, 128), (1, 41, 23, 128, 320), (2, 64, 40, 256, 256),
                                            (3, 9, 7, 64, 72)])
def test_conv3x3s2_large_tile_core(ops, monkeypatch, B, H, W, Cin, Cout):
    """The 256x256 direct-to-LDS core on the three products of the convolution, forced on at sizes
    below its usual threshold: output rows and columns that do not fill a tile, taps outside the
    image (rows fetched from the block of zeros), a contraction tail in the weight gradient."""
    monkeypatch.setenv("ATHENA_GEMM_BIG", "2")
    cd = torch.bfloat16
    g = torch.Generator().manual_seed(H * W + Cin + Cout)
    x = torch.randn(B, H, W, Cin, generator=g, dtype=torch.float64)
    w = torch.randn(3, 3, Cin, Cout, generator=g, dtype=torch.float64) * 0.1
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    dy = torch.randn(B, Ho, Wo, Cout, generator=g, dtype=torch.float64)
    xr, wr = rounded(x, cd).requires_grad_(True), rounded(w, cd).requires_grad_(True)
    ref = om.conv2d_same_s2(xr, wr)
    (ref * rounded(dy, cd)).sum().backward()
    xd, dyd = to_cd(x, cd), to_cd(dy, cd)
    w_hwio = to_cd(w, cd)
    w_t = to_cd(w.permute(3, 0, 1, 2), cd)          # [Cout][3][3][Cin]
    stats = torch.empty(2 * Cout, dtype=torch.float64, device=dev())
    out = torch.full((B, Ho, Wo, Cout), float("nan"), dtype=cd, device=dev())
    ops.conv3x3s2_fwd(xd, w_t, 1, out, B, H, W, Cin, Cout, stats)
    assert_close(out, ref.detach(), 1e-2, "conv fwd")
    o64 = out.double().cpu().reshape(-1, Cout)
    assert_close(stats[:Cout], o64.sum(0), 1e-5, "channel sums")
    assert_close(stats[Cout:], (o64 * o64).sum(0), 1e-5, "channel sums of squares")
    din = torch.full((B, H, W, Cin), float("nan"), dtype=cd, device=dev())
    ops.conv3x3s2_dgrad(dyd, w_hwio, din, B, H, W, Cin, Cout)
    assert_close(din, xr.grad, 1e-2, "conv dgrad")
    dw = torch.full((3, 3, Cin, Cout), float("nan"), dtype=torch.float32, device=dev())
    ops.conv3x3s2_wgrad(xd, dyd, dw, B, H, W, Cin, Cout)
    assert_close(dw, wr.grad, 2e-5, "conv wgrad")
    # the same entry points on the 128x128 core give the same numbers up to the order of the fp32 sums
    monkeypatch.setenv("ATHENA_GEMM_BIG", "0")
    out0 = torch.empty_like(out)
    ops.conv3x3s2_fwd(xd, w_t, 1, out0, B, H, W, Cin, Cout, stats)
    assert_close(out, out0.double(), 1e-2, "conv fwd, two cores")


@pytest.mark.parametrize("split", [True, False])
def test_gemm_wgrad_table(ops, split, monkeypatch):
    """Weight and bias gradients of many Dense layers in one launch of the 256x256 core: shapes that
    fill several tiles, shapes smaller than one, operands that are column blocks of wider matrices,
    a contraction that is not a multiple of the k-tile.  The table is far from filling the chip, so
    with ``split`` its long contractions are cut into parts (the 8000-row one into seven, the
    2000-row one into two) whose partial products a second launch sums."""
    monkeypatch.setattr(ops._WgradTables, "split_enabled", split)
    ops._WgradTables.cache.clear()
    g = torch.Generator().manual_seed(123)
    cd = torch.bfloat16
    shapes = [  # rows, M (in features), N (out features), extra columns of x / dy storage
        (1000, 256, 256, 0, 0), (130, 40, 72, 0, 8), (2000, 256, 1024, 0, 0), (777, 1024, 256, 0, 512),
        (1056, 256, 4233, 0, 7), (64, 520, 136, 24, 0), (8000, 256, 256, 512, 512)]
    problems, refs, keep = [], [], []
    for rows, M, N, xa, ya in shapes:
        ldx, ldy = (M + xa + 7) // 8 * 8, (N + ya + 7) // 8 * 8
        X = torch.randn(rows, ldx, generator=g, dtype=torch.float64)
        dY = torch.randn(rows, ldy, generator=g, dtype=torch.float64)
        xo, yo = (8 if xa >= 8 else 0), (8 if ya >= 8 else 0)      # a block that starts inside the row
        x_d, dy_d = to_cd(X, cd), to_cd(dY, cd)
        G = torch.full((M, N), float("nan"), dtype=torch.float32, device=dev())
        gb = torch.full((N,), float("nan"), dtype=torch.float32, device=dev())
        xv, yv = x_d[:, xo:xo + M], dy_d[:, yo:yo + N]
        problems.append((xv, yv, rows, M, N, ldx, ldy, G, gb if N != 72 else None))
        Xr, dYr = rounded(X, cd)[:, xo:xo + M], rounded(dY, cd)[:, yo:yo + N]
        refs.append((Xr.t() @ dYr, dYr.sum(0)))
        keep.append((x_d, dy_d))
    ops.gemm_wgrad_table(problems)
    entry = next(iter(ops._WgradTables.cache.values()))
    assert (entry[3] > len(problems) and entry[7] > 0) if split else (entry[3] == len(problems) and entry[7] == 0)
    for (xv, yv, rows, M, N, ldx, ldy, G, gb), (gr, br) in zip(problems, refs):
        assert_close(G, gr, 2e-5, "weight gradient %dx%d over %d rows" % (M, N, rows))
        if gb is not None:
            assert_close(gb, br, 2e-5, "bias gradient %d over %d rows" % (N, rows))
    # a second launch of the same table (served from the cache) reproduces the first bit for bit
    first = [p[7].clone() for p in problems]
    ops.gemm_wgrad_table(problems)
    assert all(torch.equal(a, p[7]) for a, p in zip(first, problems))
    ops._WgradTables.cache.clear()


@pytest.mark.parametrize("cd", DTYPES)
@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 21, 12, 1, 16), (1, 10, 9, 3, 40), (2, 50, 40, 1, 256)])
def test_conv_small_cin(ops, cd, B, H, W, Cin, Cout):
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(B, H, W, Cin, generator=g, dtype=torch.float64).float().double()
    w = (torch.randn(3, 3, Cin, Cout, generator=g, dtype=torch.float64) * 0.3).float().double()
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    dy = torch.randn(B, Ho, Wo, Cout, generator=g, dtype=torch.float64)
    wr = w.clone().requires_grad_(True)
    ref = om.conv2d_same_s2(x, wr)
    (ref * rounded(dy, cd)).sum().backward()
    out = torch.empty(B, Ho, Wo, Cout, dtype=cd, device=dev())
    stats = torch.empty(2 * Cout, dtype=torch.float64, device=dev())
    ops.conv_small_cin_fwd(x.float().to(dev()), w.float().to(dev()), out, B, H, W, Cin, Cout, stats)
    assert_close(out, ref.detach(), TOL[cd], "conv1 fwd")
    o64 = out.double().cpu().reshape(-1, Cout)
    assert_close(stats[:Cout], o64.sum(0), 1e-5, "channel sums")
    assert_close(stats[Cout:], (o64 * o64).sum(0), 1e-5, "channel sums of squares")
    dw = torch.empty(3, 3, Cin, Cout, dtype=torch.float32, device=dev())
    ops.conv_small_cin_wgrad(x.float().to(dev()), to_cd(dy, cd), dw, B, H, W, Cin, Cout)
    assert_close(dw, wr.grad, 2e-5, "conv1 wgrad")


@pytest.mark.parametrize("cd", DTYPES)
@pytest.mark.parametrize("rows,C", [(700, 16), (1200, 256), (333, 64)])
def test_batchnorm_relu6(ops, cd, rows, C):
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, C, generator=g, dtype=torch.float64) * 2 + 0.5
    gamma = torch.rand(C, generator=g, dtype=torch.float64) * 3 + 0.5
    beta = torch.randn(C, generator=g, dtype=torch.float64) + 1
    dy = torch.randn(rows, C, generator=g, dtype=torch.float64)
    xr = rounded(x, cd).requires_grad_(True)
    gm, bt = gamma.float().double().requires_grad_(True), beta.float().double().requires_grad_(True)
    y_bn, mean, var = om.batch_norm_train(xr.view(1, rows, 1, C), gm, bt)
    y_ref = om.relu6(y_bn).view(rows, C)
    (y_ref * rounded(dy, cd)).sum().backward()
    xd = to_cd(x, cd)
    x64 = xd.double()
    stats = torch.cat([x64.sum(0), (x64 * x64).sum(0)])
    mr = torch.empty(2 * C, dtype=torch.float32, device=dev())
    mm = torch.zeros(C, dtype=torch.float32, device=dev())
    mv = torch.ones(C, dtype=torch.float32, device=dev())
    ops.bn_finalize(stats, rows, C, 1e-3, 0.99, mr, mm, mv)
    assert_close(mr[:C], mean.detach(), 1e-5, "mean")
    assert_close(mr[C:], 1 / torch.sqrt(var.detach() + 1e-3), 1e-5, "rstd")
    assert_close(mm, 0.01 * mean.detach(), 1e-5, "moving mean")
    assert_close(mv, 0.99 + 0.01 * var.detach() * rows / (rows - 1), 1e-5, "moving variance")
    gd, bd = gamma.float().to(dev()), beta.float().to(dev())
    y = torch.empty(rows, C, dtype=cd, device=dev())
    ops.bn_relu6_apply(xd, mr, gd, bd, rows, C, y)
    assert_close(y, y_ref.detach(), TOL[cd], "bn+relu6")
    y2 = torch.empty_like(y)
    ops.bn_relu6_infer(xd, mr[:C].contiguous(), (1 / mr[C:] ** 2 - 1e-3).contiguous(), gd, bd, 1e-3, rows, C, y2)
    assert_close(y2, y_ref.detach(), TOL[cd] * 2, "bn infer")
    scratch = ops.bn_relu6_bwd_scratch(rows, C, dev())
    dx = torch.empty_like(xd)
    dg, db = torch.empty(C, device=dev()), torch.empty(C, device=dev())
    ops.bn_relu6_bwd(to_cd(dy, cd), xd, mr, gd, bd, rows, C, scratch, dx, dg, db)
    tol = 1e-4 if cd == torch.float32 else 3e-2
    assert_close(dg, gm.grad, tol, "dgamma")
    assert_close(db, bt.grad, tol, "dbeta")
    assert_close(dx, xr.grad, tol, "dx")


@pytest.mark.parametrize("d", [64, 256, 512])
def test_add_layernorm_with_the_stream_in_fp32(ops, d):
    """athena_add_layernorm_fwd_wide: an fp32 copy of ``a`` that holds the bf16 values and no dropout gives the
    plain call's z bit for bit and its y up to the rounding of z inside the statistics; with an
    unrounded ``a32`` the result follows the fp64 LayerNorm of a32 + b (reference
    athena/layers/transformer.py:305,308) more closely than the bf16 call does; the fp32 copy of y
    rounds to y."""
    rows = 37
    g = torch.Generator().manual_seed(d)
    a = (3.0 * torch.randn(rows, d, generator=g, dtype=torch.float64))
    b = torch.randn(rows, d, generator=g, dtype=torch.float64)
    gam = 1.0 + 0.2 * torch.randn(d, generator=g)
    bet = 0.1 * torch.randn(d, generator=g)
    a16, b16 = to_cd(a, torch.bfloat16), to_cd(b, torch.bfloat16)
    gd, bd = gam.to(dev()), bet.to(dev())
    z0, y0, m0, r0 = ops.add_layernorm_fwd(a16, b16, gd, bd, 1e-8, 0.0, None, 0)
    z1, y1, m1, r1, y1w = ops.add_layernorm_fwd(a16, b16, gd, bd, 1e-8, 0.0, None, 0, a32=a16.float(), want_y32=True)
    assert torch.equal(z0, z1)
    assert_close(y1, y0.float(), 1.2e-2, "y with statistics of the unrounded sum")
    assert torch.equal(y1w.to(torch.bfloat16), y1)
    z2, y2, m2, r2, y2w = ops.add_layernorm_fwd(a16, b16, gd, bd, 1e-8, 0.0, None, 0, a32=a.float().to(dev()),
                                                want_y32=True)
    zr = a + rounded(b, torch.bfloat16)
    mu = zr.mean(-1, keepdim=True)
    var = ((zr - mu) ** 2).mean(-1, keepdim=True)
    yr = (zr - mu) / torch.sqrt(var + 1e-8) * gam.double() + bet.double()
    assert_close(y2w, yr, 1e-5, "fp32 copy of y")
    assert_close(m2, mu.view(-1), 1e-5, "mean")
    err_wide = float((y2w.double().cpu() - yr).abs().max())
    zb = rounded(a, torch.bfloat16) + rounded(b, torch.bfloat16)
    err_bf16 = float((y0.double().cpu() - yr).abs().max())
    assert err_wide < 0.05 * err_bf16, (err_wide, err_bf16)
    # dropout decisions do not depend on the variant
    rng = torch.tensor([5, 3], dtype=torch.int64, device=dev())
    zd0 = ops.add_layernorm_fwd(a16, b16, gd, bd, 1e-8, 0.5, rng, 4)[0]
    zd1 = ops.add_layernorm_fwd(a16, b16, gd, bd, 1e-8, 0.5, rng, 4, a32=a16.float())[0]
    assert torch.equal(zd0, zd1)
    with pytest.raises(RuntimeError):
        ops.add_layernorm_fwd(to_cd(a[:, :d - 4], torch.bfloat16).contiguous(), None, gd[:d - 4].contiguous(),
                              bd[:d - 4].contiguous(), 1e-8, 0.0, None, 0, want_y32=True)


# ------------------------------------------------------------------------------------------ PE / embedding
@pytest.mark.parametrize("cd", DTYPES)
def test_embedding_and_pe(ops, cd):
    B, L, d, V = 3, 7, 32, 11
    g = torch.Generator().manual_seed(4)
    table = torch.randn(V, d, generator=g, dtype=torch.float64).float()
    ids = torch.randint(0, V, (B, L), generator=g).to(torch.int32)
    pe = om.make_positional_encoding(800, d)
    xs = math.sqrt(d)
    ref = table.double()[ids.long()] * xs + pe[:L].double()
    y = torch.empty(B, L, d, dtype=cd, device=dev())
    ops.embed_fwd(table.to(dev()), ids.to(dev()), pe.to(dev()), B, L, d, V, xs, 0.0, None, 0, y)
    assert_close(y, ref, TOL[cd], "embedding")
    dy = torch.randn(B, L, d, generator=g, dtype=torch.float64)
    dt_ref = torch.zeros(V, d, dtype=torch.float64)
    dt_ref.index_add_(0, ids.long().view(-1), (rounded(dy, cd) * xs).view(-1, d))
    # every row is written, rows no token points at included: what the buffer held does not matter
    dtab = torch.full((V, d), 7.0, dtype=torch.float32, device=dev())
    ops.embed_bwd(to_cd(dy, cd), ids.to(dev()), B, L, d, V, xs, 0.0, None, 0, dtab)
    assert_close(dtab, dt_ref, 1e-5, "embedding grad")
    x = torch.randn(B, L, d, generator=g, dtype=torch.float64)
    out = torch.empty(B, L, d, dtype=cd, device=dev())
    ops.add_pe_dropout(to_cd(x, cd), pe.to(dev()), B, L, d, 1.0, 0.0, None, 0, out)
    assert_close(out, rounded(x, cd) + pe[:L].double(), TOL[cd], "x + pe")
    # dropout backward with the relu6 gate
    yv = om.relu6(torch.randn(B, L, d, generator=g, dtype=torch.float64) * 4)
    dx = torch.empty(B, L, d, dtype=cd, device=dev())
    ops.dropout_bwd(to_cd(dy, cd), 1.0, 0.0, None, 0, to_cd(yv, cd), dx)
    yr = rounded(yv, cd)
    assert_close(dx, rounded(dy, cd) * ((yr > 0) & (yr < 6)).double(), 1e-6, "relu6 gate")


@pytest.mark.parametrize("d,V", [(320, 37), (600, 10), (38, 9), (530, 5)])
def test_embedding_grad_many_tokens_with_dropout(ops, d, V):
    """Thousands of tokens, ids outside the table (clamped like the forward pass clamps them), widths
    that are no multiple of a wave's 64 columns and wider than one pass over the tokens covers, a
    row count that does not fill the last workgroup, and the dropout decisions of the forward pass:
    grad = sum over tokens of dy * (y / (table * xs + pe)), i.e. the multiplier each element of the
    forward output shows.  Two runs agree bit for bit."""
    B, L = 5, 1800
    g = torch.Generator().manual_seed(11)
    ids = torch.randint(-2, V + 2, (B, L), generator=g).to(torch.int32)
    table = (torch.rand(V, d, generator=g) + 1.0)
    pe = torch.zeros(L, d)
    rng = torch.tensor([1234, 5], dtype=torch.int64, device=dev())
    y = torch.empty(B, L, d, dtype=torch.float32, device=dev())
    ops.embed_fwd(table.to(dev()), ids.to(dev()), pe.to(dev()), B, L, d, V, 2.0, 0.25, rng, 9, y)
    cl = ids.long().clamp(0, V - 1)
    mult = (y.cpu().double() / (table.double()[cl] * 2.0))
    assert bool(((mult == 0) | ((mult - 1 / 0.75).abs() < 1e-5)).all())
    mult = (mult != 0).double() / 0.75
    dy = torch.randn(B, L, d, generator=g)
    want = torch.zeros(V, d, dtype=torch.float64)
    want.index_add_(0, cl.view(-1), (dy.double() * mult * 2.0).view(-1, d))
    got = [torch.full((V, d), float("nan"), dtype=torch.float32, device=dev()) for _ in range(2)]
    for t in got:
        ops.embed_bwd(dy.to(dev()), ids.to(dev()), B, L, d, V, 2.0, 0.25, rng, 9, t)
    assert_close(got[0], want, 1e-5, "embedding grad, many tokens")
    assert torch.equal(got[0], got[1])


def test_weighted_total_of_loss_terms(ops):
    """mtl_weight * main + (1 - mtl_weight) * extra (reference athena/models/mtl_seq2seq.py:95) as one
    launch, and the gradients it hands its terms: inside a solver step constants, outside products
    with whatever arrives."""
    from athena_amd.loss import weighted_total
    a = torch.tensor(3.25, device=dev(), requires_grad=True)
    b = torch.tensor(-1.5, device=dev(), requires_grad=True)
    for w in (0.5, 0.3, 1.0, 0.0):
        total = weighted_total([a, b], [w, 1.0 - w])
        assert abs(float(total) - (w * 3.25 + (1.0 - w) * -1.5)) < 1e-6
        ga, gb = torch.autograd.grad(total * 2.0, [a, b])
        assert abs(float(ga) - 2.0 * w) < 1e-7 and abs(float(gb) - 2.0 * (1.0 - w)) < 1e-7
        with ops.bf16_grad_handover():
            total = weighted_total([a, b], [w, 1.0 - w])
            ga, gb = torch.autograd.grad(total, [a, b], ops.const_scalar(total.device, 1.0))
        assert float(ga) == np.float32(w) and float(gb) == np.float32(1.0 - w)
    out = torch.empty(1, device=dev())
    terms = [torch.tensor([float(i)], device=dev()) for i in range(8)]
    ops.weighted_sum(terms, [0.5] * 8, out)
    assert float(out) == 14.0
    with pytest.raises(RuntimeError):
        ops.weighted_sum(terms + terms[:1], [0.5] * 9, out)


# ------------------------------------------------------------------------------------------ CTC
def _ctc_case(B, T, V, U, seed, ragged=True, repeats=True):
    rng = np.random.default_rng(seed)
    logits = rng.normal(size=(B, T, V)) * 2.0
    labels = rng.integers(1, V - 1, size=(B, U)).astype(np.int32)
    if repeats and U > 2:
        labels[0, 1] = labels[0, 0]
        labels[-1, -1] = labels[-1, -2]
    if ragged:
        llen = rng.integers(max(1, U // 2), U + 1, size=B).astype(np.int32)
        llen[0] = U
        tlen = rng.integers(min(T, 2 * U + 2), T + 1, size=B).astype(np.int32)
        tlen[0] = T
    else:
        llen = np.full(B, U, np.int32)
        tlen = np.full(B, T, np.int32)
    for b in range(B):
        labels[b, llen[b]:] = 0
    return logits, labels, llen, tlen


@pytest.mark.parametrize("B,T,V,U", [(3, 12, 7, 4), (4, 40, 30, 9), (2, 150, 50, 40), (5, 64, 4233, 20),
                                     (2, 90, 101, 33), (2, 300, 60, 100), (1, 250, 40, 200), (2, 30, 12, 0)])
def test_ctc_loss_and_grad_fp32(ops, B, T, V, U):
    Ue = max(U, 1)
    logits, labels, llen, tlen = _ctc_case(B, T, V, Ue, seed=B * 1000 + T + U)
    if U == 0:
        llen[:] = 0
        labels[:] = 0
    lt = torch.from_numpy(logits).float()
    loss_ref, grad_ref = octc.ctc_loss_textbook(lt.double().numpy(), labels, llen, tlen, blank_index=-1)
    ldl = (V + 7) // 8 * 8
    buf = torch.zeros(B, T, ldl, dtype=torch.float32, device=dev())
    buf[..., :V] = lt.to(dev())
    ws = torch.empty(ops.ctc_workspace_bytes(B, T, Ue), dtype=torch.uint8, device=dev())
    per = torch.empty(B, dtype=torch.float32, device=dev())
    lab_d = torch.from_numpy(labels).to(dev())
    ll_d, tl_d = torch.from_numpy(llen).to(dev()), torch.from_numpy(tlen).to(dev())
    best = torch.full((B * T,), -7, dtype=torch.int32, device=dev())
    ops.ctc_loss_fwd(buf, ldl, lab_d, ll_d, tl_d, B, T, V, Ue, V - 1, per, ws, argmax=best)
    assert_close(per, loss_ref, 1e-5, "ctc loss")
    # the per-frame arg-max that the same pass leaves for the accuracy metric: exact inside
    # logit_length, zero beyond
    want = lt.argmax(-1)
    got = best.view(B, T).cpu()
    for b in range(B):
        assert got[b, :tlen[b]].tolist() == want[b, :tlen[b]].tolist()
        assert (got[b, tlen[b]:] == 0).all()
    grad = torch.full((B, T, ldl), float("nan"), dtype=torch.float32, device=dev())
    gs = torch.tensor([0.5], dtype=torch.float32, device=dev())
    ops.ctc_loss_bwd(buf, ldl, lab_d, ll_d, tl_d, B, T, V, Ue, V - 1, gs, 2.0 / B, grad, ws)
    assert_close(grad[..., :V], grad_ref / B, 1e-4, "ctc grad")
    assert float(grad[..., V:].abs().max()) == 0.0 if ldl > V else True
    for b in range(B):
        assert float(grad[b, tlen[b]:].abs().max()) == 0.0 if tlen[b] < T else True


def test_ctc_matches_brute_force(ops):
    """Known answers: every alignment enumerated (tests/golden/ctc_kat.json)."""
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ctc_kat.json")))
    for case in kat["cases"]:
        logits = torch.tensor(case["logits"], dtype=torch.float32)
        B, T, V = logits.shape
        labels = torch.tensor(case["labels"], dtype=torch.int32)
        U = labels.shape[1]
        ldl = 8
        buf = torch.zeros(B, T, ldl, device=dev())
        buf[..., :V] = logits.to(dev())
        ws = torch.empty(ops.ctc_workspace_bytes(B, T, U), dtype=torch.uint8, device=dev())
        per = torch.empty(B, device=dev())
        ops.ctc_loss_fwd(buf, ldl, labels.to(dev()), torch.tensor(case["label_length"], dtype=torch.int32).to(dev()),
                         torch.tensor(case["logit_length"], dtype=torch.int32).to(dev()), B, T, V, U, V - 1, per, ws)
        assert_close(per, torch.tensor(case["loss"]), 1e-5, "brute-force KAT")


def _run_ctc(ops, logits, labels, llen, tlen, gscale=1.0):
    """(per-sample loss, gradient [B,T,V]) of the HIP kernels on fp32 logits [B,T,V]."""
    B, T, V = logits.shape
    U = labels.shape[1]
    ldl = (V + 7) // 8 * 8
    buf = torch.zeros(B, T, ldl, dtype=torch.float32, device=dev())
    buf[..., :V] = torch.as_tensor(logits).float().to(dev())
    ws = torch.empty(ops.ctc_workspace_bytes(B, T, U), dtype=torch.uint8, device=dev())
    per = torch.empty(B, dtype=torch.float32, device=dev())
    lab_d = torch.as_tensor(labels, dtype=torch.int32).to(dev())
    ll_d = torch.as_tensor(llen, dtype=torch.int32).to(dev())
    tl_d = torch.as_tensor(tlen, dtype=torch.int32).to(dev())
    ops.ctc_loss_fwd(buf, ldl, lab_d, ll_d, tl_d, B, T, V, U, V - 1, per, ws)
    grad = torch.full((B, T, ldl), float("nan"), dtype=torch.float32, device=dev())
    ops.ctc_loss_bwd(buf, ldl, lab_d, ll_d, tl_d, B, T, V, U, V - 1, None, gscale, grad, ws)
    return per, grad[..., :V]


def test_ctc_infeasible_matches_tf_dense(ops):
    """Samples whose labels do not fit their frames: tf.nn.ctc_loss's dense branch starts and ends
    its state vectors at log(1e-307) instead of log 0, so it returns a finite loss of about 7e2 and
    a gradient along the leaked paths.  Loss and gradient must equal the restatement of that
    branch (oracle.ctc.ctc_loss_tf_dense), also when feasible and infeasible samples share a batch
    and when a sample has no frames at all."""
    rng = np.random.default_rng(11)
    B, T, V, U = 6, 7, 6, 5
    logits = np.float32(rng.normal(size=(B, T, V)) * 1.5)
    labels = np.array([[1, 2, 3, 1, 2],      # 5 labels, 3 frames
                       [2, 2, 2, 0, 0],      # 3 equal labels need 5 frames, has 4
                       [1, 3, 0, 0, 0],      # fits
                       [4, 4, 1, 1, 0],      # 4 labels with two repeats need 6 frames, has 5
                       [3, 1, 2, 0, 0],      # no frames at all
                       [0, 0, 0, 0, 0]],     # no labels, no frames
                      dtype=np.int32)
    llen = np.array([5, 3, 2, 4, 3, 0], np.int32)
    tlen = np.array([3, 4, 7, 5, 0, 0], np.int32)
    loss_ref, grad_ref = octc.ctc_loss_tf_dense(logits.astype(np.float64), labels, llen, tlen, blank_index=-1)
    assert loss_ref[0] > 500 and loss_ref[1] > 500 and loss_ref[2] < 50 and loss_ref[3] > 500
    per, grad = _run_ctc(ops, logits, labels, llen, tlen)
    assert_close(per, loss_ref, 1e-5, "ctc loss, TF dense semantics")
    for b in range(B):
        if tlen[b] > 0:
            assert_close(grad[b], grad_ref[b], 1e-4, "ctc grad, TF dense semantics, sample %d" % b)
        else:
            assert float(grad[b].abs().max()) == 0.0
    # a longer infeasible lattice (several states per lane, many steps)
    B, T, V, U = 2, 60, 40, 70
    logits = np.float32(rng.normal(size=(B, T, V)) * 2.0)
    labels = rng.integers(1, V - 1, size=(B, U)).astype(np.int32)
    llen = np.array([70, 20], np.int32)
    labels[1, 20:] = 0
    tlen = np.array([60, 60], np.int32)
    loss_ref, grad_ref = octc.ctc_loss_tf_dense(logits.astype(np.float64), labels, llen, tlen, blank_index=-1)
    assert loss_ref[0] > 500 and loss_ref[1] < 500
    per, grad = _run_ctc(ops, logits, labels, llen, tlen)
    assert_close(per, loss_ref, 1e-5, "ctc loss, TF dense semantics (long)")
    for b in range(B):
        assert_close(grad[b], grad_ref[b], 1e-4, "ctc grad, TF dense semantics (long), sample %d" % b)


def _run_ctc_fused(ops, logits, labels, llen, tlen, gscale=1.0, grad_dtype=torch.float32):
    """(per-sample loss, gradient [B,T,V], arg-max) of the one-pass entry point on fp32 logits [B,T,V]."""
    B, T, V = logits.shape
    U = labels.shape[1]
    ldl = (V + 7) // 8 * 8
    buf = torch.zeros(B, T, ldl, dtype=torch.float32, device=dev())
    buf[..., :V] = torch.as_tensor(logits).float().to(dev())
    ws = torch.empty(ops.ctc_workspace_bytes(B, T, U), dtype=torch.uint8, device=dev())
    per = torch.empty(B, dtype=torch.float32, device=dev())
    lab_d = torch.as_tensor(labels, dtype=torch.int32).to(dev())
    ll_d = torch.as_tensor(llen, dtype=torch.int32).to(dev())
    tl_d = torch.as_tensor(tlen, dtype=torch.int32).to(dev())
    grad = torch.full((B, T, ldl), float("nan"), dtype=grad_dtype, device=dev())
    best = torch.full((B * T,), -7, dtype=torch.int32, device=dev())
    ops.ctc_loss_fwd_bwd(buf, ldl, lab_d, ll_d, tl_d, B, T, V, U, V - 1, per, ws, gscale, grad, argmax=best)
    if ldl > V:
        assert float(grad[..., V:].float().abs().max()) == 0.0
    return per, grad[..., :V], best.view(B, T)


@pytest.mark.parametrize("B,T,V,U", [(3, 12, 7, 4), (4, 40, 30, 9), (2, 150, 50, 40), (5, 64, 4233, 20),
                                     (2, 90, 101, 33), (2, 300, 60, 100), (1, 250, 40, 200), (2, 30, 12, 0),
                                     (2, 20, 9000, 6), (2, 1, 5, 1), (3, 2, 6, 1),
                                     # every lattice size the library is built for: 2U+1 states on 64 lanes
                                     (1, 320, 30, 150), (2, 650, 20, 300), (1, 950, 20, 450), (1, 1030, 12, 511)])
def test_ctc_one_pass_loss_and_grad(ops, B, T, V, U):
    """athena_ctc_loss_fwd_bwd (loss and gradient from one pass over the logits) against the textbook
    recursion, with repeated labels (several lattice states sharing a logits column), ragged lengths,
    rows wider than the register path, and against the two-call path."""
    Ue = max(U, 1)
    logits, labels, llen, tlen = _ctc_case(B, T, V, Ue, seed=B * 1000 + T + U)
    if U == 0:
        llen[:] = 0
        labels[:] = 0
    if U >= 4:
        labels[0, 3] = labels[0, 0]          # a repeat that is not adjacent
    lt = torch.from_numpy(logits).float()
    loss_ref, grad_ref = octc.ctc_loss_textbook(lt.double().numpy(), labels, llen, tlen, blank_index=-1)
    per, grad, best = _run_ctc_fused(ops, lt, labels, llen, tlen, gscale=0.7 / B)
    assert_close(per, loss_ref, 1e-5, "ctc loss (one pass)")
    assert_close(grad, 0.7 * grad_ref / B, 1e-4, "ctc grad (one pass)")
    want = lt.argmax(-1)
    for b in range(B):
        assert best[b, :tlen[b]].cpu().tolist() == want[b, :tlen[b]].tolist()
        if tlen[b] < T:
            assert float(grad[b, tlen[b]:].abs().max()) == 0.0
    per2, grad2 = _run_ctc(ops, lt, labels, llen, tlen, gscale=0.7 / B)
    assert torch.equal(per, per2)
    assert_close(grad, grad2.double(), 1e-6, "one pass against two calls")
    # bf16 gradient: the same values rounded once
    _, g16, _ = _run_ctc_fused(ops, lt, labels, llen, tlen, gscale=0.7 / B, grad_dtype=torch.bfloat16)
    assert_close(g16.float(), grad.to(torch.bfloat16).float().double(), 1e-2, "bf16 gradient (one pass)")


def test_ctc_one_pass_infeasible_matches_tf_dense(ops):
    """Samples without a fitting alignment through the one-pass entry point: TensorFlow's dense-branch
    loss and gradient (the slow path writes the gradient columns itself)."""
    rng = np.random.default_rng(11)
    B, T, V, U = 6, 7, 6, 5
    logits = np.float32(rng.normal(size=(B, T, V)) * 1.5)
    labels = np.array([[1, 2, 3, 1, 2], [2, 2, 2, 0, 0], [1, 3, 0, 0, 0], [4, 4, 1, 1, 0], [3, 1, 2, 0, 0],
                       [0, 0, 0, 0, 0]], dtype=np.int32)
    llen = np.array([5, 3, 2, 4, 3, 0], np.int32)
    tlen = np.array([3, 4, 7, 5, 0, 0], np.int32)
    loss_ref, grad_ref = octc.ctc_loss_tf_dense(logits.astype(np.float64), labels, llen, tlen, blank_index=-1)
    per, grad, _ = _run_ctc_fused(ops, logits, labels, llen, tlen)
    assert_close(per, loss_ref, 1e-5, "ctc loss, TF dense semantics (one pass)")
    for b in range(B):
        if tlen[b] > 0:
            assert_close(grad[b], grad_ref[b], 1e-4, "ctc grad, TF dense semantics (one pass), sample %d" % b)
        else:
            assert float(grad[b].abs().max()) == 0.0
    B, T, V, U = 2, 60, 40, 70
    logits = np.float32(rng.normal(size=(B, T, V)) * 2.0)
    labels = rng.integers(1, V - 1, size=(B, U)).astype(np.int32)
    llen = np.array([70, 20], np.int32)
    labels[1, 20:] = 0
    tlen = np.array([60, 60], np.int32)
    loss_ref, grad_ref = octc.ctc_loss_tf_dense(logits.astype(np.float64), labels, llen, tlen, blank_index=-1)
    per, grad, _ = _run_ctc_fused(ops, logits, labels, llen, tlen)
    assert_close(per, loss_ref, 1e-5, "ctc loss, TF dense semantics (one pass, long)")
    for b in range(B):
        assert_close(grad[b], grad_ref[b], 1e-4, "ctc grad, TF dense semantics (one pass, long), sample %d" % b)


def test_ctc_extreme_logits_fall_back_to_the_log_domain(ops):
    """Emissions beyond what the probability-domain recursion holds (a label 400 nats below blank on
    the only path): the sample is redone on logarithms and still matches the oracle."""
    rng = np.random.default_rng(3)
    B, T, V, U = 2, 6, 5, 3
    logits = np.float32(rng.normal(size=(B, T, V)))
    logits[0, :, 1] -= 400.0
    labels = np.array([[1, 2, 1], [2, 3, 0]], np.int32)
    llen = np.array([3, 2], np.int32)
    tlen = np.array([6, 6], np.int32)
    loss_ref, grad_ref = octc.ctc_loss_tf_dense(logits.astype(np.float64), labels, llen, tlen, blank_index=-1)
    per, grad, _ = _run_ctc_fused(ops, logits, labels, llen, tlen)
    assert torch.isfinite(per).all()
    assert_close(per, loss_ref, 1e-5, "ctc loss with extreme logits")
    assert_close(grad, grad_ref, 1e-4, "ctc grad with extreme logits")


@pytest.mark.parametrize("where", ["end", "start", "both"])
def test_ctc_follows_tensorflow_where_its_finite_log_zero_shows(ops, where):
    """TensorFlow's dense branch starts and ends its recursion on a finite log-zero, log(1e-307) =
    -706.9, for the states an alignment may not start or end in, and reads the loss from the backward
    scan.  When the last frame puts e^-708 on both legal end states (the last label and the blank) and
    ordinary mass on the others, alignments that end in a wrong state are likelier than the fitting
    ones and leak into its loss and gradient; the same at the first frame changes its gradient only.
    The textbook recursion (true zeros) gives another value there.  The kernels notice from their last
    alpha / beta vectors that the leak would show and hand the utterance to the path that restates
    TensorFlow's recursion: loss and gradient follow oracle.ctc.ctc_loss_tf_dense, and the test
    asserts that the two statements really differ on this input."""
    rng = np.random.default_rng(11)
    B, T, V, U = 2, 12, 6, 3
    logits = np.float32(rng.normal(size=(B, T, V)))
    labels = np.array([[1, 2, 3], [2, 4, 1]], np.int32)
    if where in ("end", "both"):
        logits[0, T - 1, V - 1] = -708.0       # the blank and the last label of utterance 0, last frame
        logits[0, T - 1, 3] = -708.0
    if where in ("start", "both"):
        logits[0, 0, V - 1] = -708.0           # the blank and its first label, first frame
        logits[0, 0, 1] = -708.0
    llen = np.array([3, 3], np.int32)
    tlen = np.array([T, T], np.int32)
    l64 = logits.astype(np.float64)
    loss_tf, grad_tf = octc.ctc_loss_tf_dense(l64, labels, llen, tlen, blank_index=-1)
    loss_tb, grad_tb = octc.ctc_loss_textbook(l64, labels, llen, tlen, blank_index=-1)
    assert loss_tb[0] > 707.0
    if where != "start":
        assert loss_tb[0] - loss_tf[0] > 0.2                               # the two statements differ
    assert np.abs(grad_tf[0] - grad_tb[0]).max() > 0.1
    assert abs(loss_tf[1] - loss_tb[1]) < 1e-9 * loss_tb[1]                # an ordinary utterance: they agree
    per, grad, _ = _run_ctc_fused(ops, logits, labels, llen, tlen)
    assert torch.isfinite(grad).all()
    assert_close(per, loss_tf, 1e-5, "ctc loss where the finite log-zero shows")
    assert_close(grad, grad_tf, 1e-4, "ctc gradient where the finite log-zero shows")
    per2, grad2 = _run_ctc(ops, logits, labels, llen, tlen)
    assert_close(per2, loss_tf, 1e-5, "ctc loss where the finite log-zero shows (two calls)")
    assert_close(grad2, grad_tf, 1e-4, "ctc gradient where the finite log-zero shows (two calls)")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_rescale_unless(ops, dtype):
    x = torch.randn(5, 1000, device=dev()).to(dtype)
    keep = x.clone()
    up = torch.tensor([0.5], dtype=torch.float32, device=dev())
    ops.rescale_unless(x, up, 0.5)
    assert torch.equal(x, keep)
    ops.rescale_unless(x, up, 0.25)
    assert_close(x.float(), (keep.float() * 2.0).double(), 1e-6, "rescale")


def test_ctc_loss_function_with_expected_upstream():
    """CTCLoss(upstream=w): the gradient written in the forward pass equals the two-call gradient, and
    an upstream factor other than the expected one is corrected in the backward pass."""
    from athena_amd.loss import CTCLoss
    B, T, V, U = 3, 30, 20, 6
    logits, labels, llen, tlen = _ctc_case(B, T, V, U, seed=77)
    samples = {"output": torch.from_numpy(labels).to(dev()), "output_length": torch.from_numpy(llen).to(dev())}
    tl = torch.from_numpy(tlen).to(dev())
    grads = []
    for upstream, factor in ((None, 0.3), (0.3, 0.3), (0.3, 1.1)):
        x = torch.from_numpy(logits).float().to(dev()).requires_grad_(True)
        loss = CTCLoss(blank_index=-1)(x, samples, tl, upstream=upstream)
        (loss * factor).backward()
        grads.append((float(loss), x.grad.clone()))
    assert grads[0][0] == grads[1][0] == grads[2][0]
    assert_close(grads[1][1], grads[0][1].double(), 1e-6, "gradient written with the loss")
    assert_close(grads[2][1], grads[0][1].double() * (1.1 / 0.3), 1e-6, "gradient corrected in the backward pass")
    # a second backward pass over a retained graph gets a gradient of its own, not the first one
    # scaled twice: x.grad accumulates 0.3 g + 0.7 g
    x = torch.from_numpy(logits).float().to(dev()).requires_grad_(True)
    loss = CTCLoss(blank_index=-1)(x, samples, tl, upstream=0.3)
    (loss * 0.3).backward(retain_graph=True)
    (loss * 0.7).backward()
    assert_close(x.grad, grads[0][1].double() / 0.3, 1e-6, "two backward passes over one forward pass")


def test_ctc_label_row_too_long_is_refused(ops):
    """Label rows beyond 2U+1 = 1024 lattice states are refused with a clear error, not truncated."""
    B, T, V, U = 1, 8, 16, 600
    buf = torch.zeros(B, T, V, device=dev())
    assert ops.ctc_workspace_bytes(B, T, 511) > 0
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev())
    per = torch.empty(B, device=dev())
    with pytest.raises(RuntimeError, match="too long"):
        ops.ctc_loss_fwd(buf, V, torch.ones(B, U, dtype=torch.int32, device=dev()),
                         torch.tensor([U], dtype=torch.int32).to(dev()), torch.tensor([T], dtype=torch.int32).to(dev()),
                         B, T, V, U, V - 1, per, ws)


# ------------------------------------------------------------------------------------------ CE / argmax
@pytest.mark.parametrize("ls", [0.0, 0.1])
def test_ce_loss(ops, ls):
    B, L, V = 4, 9, 53
    g = torch.Generator().manual_seed(8)
    logits = (torch.randn(B, L, V, generator=g, dtype=torch.float64) * 3).float()
    tg = torch.randint(1, V, (B, L), generator=g).to(torch.int32)
    tg[1, 5:] = 0
    tg[3, 2:] = 0
    lr = logits.double().requires_grad_(True)
    y = F.one_hot(tg.long(), V).double()
    if ls:
        y = (1 - ls) * y + ls / V
    rl = -(y * F.log_softmax(lr, -1)).sum(-1) * (tg != 0).double()
    (rl.sum(-1).mean() * 0.7).backward()
    ldl = 56
    buf = torch.zeros(B, L, ldl, device=dev())
    buf[..., :V] = logits.to(dev())
    row_loss = torch.empty(B * L, device=dev())
    lse = torch.empty(B * L, device=dev())
    ops.ce_loss_fwd(buf, ldl, tg.to(dev()).view(-1), B * L, V, ls, row_loss, lse)
    assert_close(row_loss, rl.detach().view(-1), 1e-5, "ce row loss")
    grad = torch.full((B, L, ldl), float("nan"), device=dev())
    gs = torch.tensor([0.7], device=dev())
    ops.ce_loss_bwd(buf, ldl, tg.to(dev()).view(-1), B * L, V, ls, lse, gs, 1.0 / B, grad)
    assert_close(grad[..., :V], lr.grad, 1e-4, "ce grad")
    assert float(grad[..., V:].abs().max()) == 0.0
    out = torch.empty(1, device=dev())
    ops.group_sum(row_loss, 1, B * L, 1.0 / B, out)
    assert_close(out, rl.detach().sum(-1).mean().view(1), 1e-5, "mean over batch of sums")


def test_argmax_rows(ops):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(50, 4240, generator=g)
    x[3, 7] = x[3, 4000] = 100.0          # first maximum wins
    out = torch.empty(50, dtype=torch.int32, device=dev())
    ops.argmax_rows(x.to(dev()), 4240, 50, 4233, out)
    assert out.cpu().tolist() == x[:, :4233].argmax(-1).tolist()


# ------------------------------------------------------------------------------------------ optimizer
def test_clip_adam_and_shadows(ops):
    from athena_amd import _lib
    g = torch.Generator().manual_seed(6)
    sizes = [64 * 3, 64 * 70, 64, 64 * 130]
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    p0 = torch.randn(n, generator=g)
    gr = torch.randn(n, generator=g) * torch.tensor(np.repeat([0.01, 5.0, 1.0, 0.5], sizes)).float()
    m0 = torch.randn(n, generator=g) * 0.1
    v0 = torch.rand(n, generator=g) * 0.1
    cfgd = dict(d_model=256, warmup_steps=10, k=1.0, decay_steps=4, decay_rate=0.5)
    for step in (0, 7):
        p_ref, m_ref, v_ref = p0.double().clone(), m0.double().clone(), v0.double().clone()
        lr = om.warmup_lr(step, **cfgd)
        for i in range(len(sizes)):
            sl = slice(int(offs[i]), int(offs[i + 1]))
            gg = om.clip_by_norm(gr.double()[sl] * 0.5, 3.0)
            om.adam_update(p_ref[sl], gg, m_ref[sl], v_ref[sl], step, lr)
        p, m, v = p0.to(dev()), m0.to(dev()), v0.to(dev())
        st = torch.tensor([step], dtype=torch.int64, device=dev())
        norms = torch.empty(len(sizes), device=dev())
        lr_out = torch.empty(1, device=dev())
        cfg = _lib.AdamCfg(0.9, 0.999, 1e-7, 3.0, 256.0, 10.0, 1.0, 4.0, 0.5, 0.5, 0.0, 0.0, 0.0, 0.0)
        ops.clip_adam_step(p, gr.to(dev()), m, v, torch.from_numpy(offs).to(dev()), len(sizes), norms, st, cfg, lr_out)
        assert int(st.item()) == step + 1
        assert_close(lr_out, torch.tensor([lr]), 1e-5, "lr")
        assert_close(m, m_ref, 1e-5, "m")
        assert_close(v, v_ref, 1e-5, "v")
        assert_close(p, p_ref, 1e-5, "param")


def test_expdecay_adam_schedule_and_update():
    """ExponentialDecayAdam (reference athena/utils/learning_rate.py:89-146): the learning rate stays at
    initial_lr until start_decay_steps, then halves every decay_steps down to final_lr; the update is
    Keras Adam with epsilon 1e-6 through the same fused kernel."""
    from athena_amd import ops
    from athena_amd.runtime import ParamStore, glorot_uniform
    from athena_amd.utils.learning_rate import ExponentialDecayAdam
    cfg = {"initial_lr": 0.01, "decay_steps": 3, "decay_rate": 0.5, "start_decay_steps": 4, "final_lr": 0.002}
    store = ParamStore(torch.float32, device=dev())
    store.register("w", (16, 24), glorot_uniform(16, 24))
    store.register("b", (24,), glorot_uniform(24, 24))
    store.finalize(seed=2)
    opt = ExponentialDecayAdam(dict(cfg))
    opt.attach(store)
    g = torch.Generator().manual_seed(9)
    p_ref = store.params[:store.n_train].double().cpu().clone()
    m_ref, v_ref = torch.zeros_like(p_ref), torch.zeros_like(p_ref)
    want_lr = [0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.01, 0.005, 0.005, 0.005, 0.0025, 0.0025, 0.0025, 0.002, 0.002]
    for step, lr in enumerate(want_lr):
        assert abs(opt.lr(step) - lr) < 1e-12
        grad = torch.randn(store.n_train, generator=g)
        store.grads.copy_(grad.to(dev()))
        opt.apply_gradients(clip_norm=0.0)
        assert abs(float(opt.lr_t) - lr) < 1e-7 * max(1.0, lr / 1e-3), (step, float(opt.lr_t), lr)
        om.adam_update(p_ref, grad.double(), m_ref, v_ref, step, lr, eps=1e-6)
    assert opt.iterations == len(want_lr)
    assert_close(store.params[:store.n_train], p_ref, 1e-5, "parameters after %d updates" % len(want_lr))


def _encoder_layer_run(monkeypatch, fused, dff, B, T, rate, seed=5):
    """One forward + backward pass of a bf16 encoder layer (d_model 256) on seeded inputs; returns the
    output, the input gradient and a copy of all parameter gradients."""
    from athena_amd.layers.commons import bind_all
    from athena_amd.layers.transformer import TransformerEncoderLayer
    from athena_amd.runtime import ParamStore
    monkeypatch.setenv("ATHENA_LAYER_TAIL", "1" if fused else "0")
    store = ParamStore(torch.bfloat16, device=dev())
    layer = TransformerEncoderLayer(store, "enc", 256, 4, dff, rate)
    store.finalize(seed=seed)
    bind_all(layer)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        # LayerNorm parameters and biases away from their 1 / 0 initial values
        for name, p_ in layer.named_parameters():
            if p_.dim() == 1:
                p_.add_(0.3 * torch.randn(p_.shape, generator=g).to(p_.device))
    store.refresh_shadows(force=True)
    x = torch.randn(B, T, 256, generator=g).to(dev()).to(torch.bfloat16).requires_grad_(True)
    dy = torch.randn(B, T, 256, generator=g).to(dev()).to(torch.bfloat16)
    store.grads.zero_()
    y = layer(x, None, training=True)
    y.backward(dy)
    torch.cuda.synchronize()
    grads = {n: p_.grad.detach().float().clone() for n, p_ in layer.named_parameters()}
    return y.detach().float(), x.grad.detach().float(), grads


@pytest.fixture(params=["2048", "0"], ids=["rows16", "rows32"])
def tail_rows(request, monkeypatch):
    """Both builds of the one-launch layer tails: workgroups of 16 rows (what a launch with few rows
    gets) and of 32 rows (the tall launches), whatever the row count of the test."""
    monkeypatch.setenv("ATHENA_TAIL_SMALL_ROWS", request.param)
    return request.param


@pytest.mark.parametrize("dff,B,T,rate", [(1024, 4, 64, 0.0), (512, 3, 50, 0.1), (256, 1, 7, 0.1)])
def test_layer_tail_matches_separate_kernels(ops, monkeypatch, tail_rows, dff, B, T, rate):
    """The one-launch layer tail (output projection, LayerNorms, feed-forward block; reference
    athena/layers/transformer.py:296-310) against the same layer run as separate Dense / LayerNorm
    launches: same dropout decisions, so the two differ by bf16 rounding of intermediates only (the
    fused kernel keeps them in fp32 where the separate kernels store bf16).  Row counts that do not
    fill the last 32-row block are included."""
    y0, dx0, g0 = _encoder_layer_run(monkeypatch, False, dff, B, T, rate)
    y1, dx1, g1 = _encoder_layer_run(monkeypatch, True, dff, B, T, rate)
    assert torch.isfinite(y1).all() and torch.isfinite(dx1).all()
    assert rel_l2(y1, y0) < 6e-3, ("y", rel_l2(y1, y0))
    assert rel_l2(dx1, dx0) < 1.5e-2, ("dx", rel_l2(dx1, dx0))
    for name in g0:
        assert torch.isfinite(g1[name]).all(), name
        if name.endswith("wk.bias"):
            continue        # exactly zero in exact arithmetic (a shift of all keys cancels in the softmax)
        assert rel_l2(g1[name], g0[name]) < 1.5e-2, (name, rel_l2(g1[name], g0[name]))


@pytest.mark.parametrize("dff", [1024, 0])
def test_layer_tail_fp32_residual_hand_over(ops, monkeypatch, tail_rows, dff):
    """The fp32 copy of the residual stream that one tail leaves for the next (``y32`` / ``x32``): an fp32
    copy that holds exactly the bf16 values changes nothing, bit for bit; the copy a launch leaves rounds
    to the bf16 output it writes; and a residual input that differs from its bf16 rounding shows in the
    output by about that difference (reference athena/layers/transformer.py:296-310 keeps the stream in
    fp32 throughout).  With and without the feed-forward part, rows that do not fill the last block."""
    from athena_amd.layers.commons import bind_all
    from athena_amd.layers.transformer import TransformerEncoderLayer
    from athena_amd.runtime import ParamStore
    store = ParamStore(torch.bfloat16, device=dev())
    layer = TransformerEncoderLayer(store, "enc", 256, 4, max(dff, 256), 0.1)
    store.finalize(seed=3)
    bind_all(layer)
    store.refresh_shadows(force=True)
    rows = 3 * 50 + 7
    g = torch.Generator().manual_seed(8)
    c = torch.randn(rows, 256, generator=g).to(dev()).to(torch.bfloat16)
    x = (2.0 * torch.randn(rows, 256, generator=g)).to(dev())
    xb = x.to(torch.bfloat16)
    rng = torch.tensor([7, 1], dtype=torch.int64, device=dev())

    def run(x32, want):
        return ops.layer_tail_fwd(c, xb, layer.self_attn.dense, layer.norm1, layer.ffn0 if dff else None,
                                  layer.ffn1 if dff else None, layer.norm2 if dff else None, 1e-8, 0.1, rng,
                                  layer.site_res1, layer.site_ffn_h if dff else 0, layer.site_res2 if dff else 0,
                                  x32=x32, want_y32=want)
    last = "y2" if dff else "y1"
    plain = run(None, True)
    same = run(xb.float(), True)
    for k in plain:
        assert torch.equal(plain[k], same[k]), k
    assert torch.equal(plain["y32"].to(torch.bfloat16), plain[last])
    assert torch.isfinite(plain["y32"]).all()
    wide = run(x, False)
    # z1 = x + dropout(projection): the stored copy is the rounding of the sum with the unrounded x
    diff = (wide["z1"].float() - plain["z1"].float()).abs().max()
    assert 0 < float(diff) <= 2 * float((x - xb.float()).abs().max()) + 2 ** -6 * float(x.abs().max())
    assert rel_l2(wide[last].float(), plain[last].float()) < 1e-2


@pytest.mark.parametrize("dff,B,T", [(1024, 4, 64), (512, 2, 45)])
def test_layer_tail_against_the_oracle_layer(ops, monkeypatch, tail_rows, dff, B, T):
    """The bf16 encoder layer with the one-launch tails (the benchmarked path) against
    ``oracle.model.encoder_layer`` (TransformerEncoderLayer.call, reference
    athena/layers/transformer.py:296-310) in fp64 on the values the GPU sees -- input and weight
    matrices rounded to bf16, vectors in fp32 -- without dropout: output, input gradient and every
    parameter gradient within 1e-2 in relative L2."""
    from athena_amd.layers.commons import bind_all
    from athena_amd.layers.transformer import TransformerEncoderLayer
    from athena_amd.runtime import ParamStore
    monkeypatch.setenv("ATHENA_LAYER_TAIL", "1")
    store = ParamStore(torch.bfloat16, device=dev())
    layer = TransformerEncoderLayer(store, "enc", 256, 4, dff, 0.0)
    store.finalize(seed=21)
    bind_all(layer)
    g = torch.Generator().manual_seed(21)
    with torch.no_grad():
        for name, p_ in layer.named_parameters():
            if p_.dim() == 1:
                p_.add_(0.3 * torch.randn(p_.shape, generator=g).to(p_.device))
    store.refresh_shadows(force=True)
    x = torch.randn(B, T, 256, generator=g).to(torch.bfloat16)
    dy = torch.randn(B, T, 256, generator=g).to(torch.bfloat16)
    xg = x.to(dev()).requires_grad_(True)
    store.grads.zero_()
    calls = []
    real = ops.layer_tail_fwd
    monkeypatch.setattr(ops, "layer_tail_fwd", lambda *a, **k: calls.append(1) or real(*a, **k))
    y = layer(xg, None, training=True)
    y.backward(dy.to(dev()))
    torch.cuda.synchronize()
    assert calls, "the layer did not take the one-launch tail"
    got = {n: p_.grad.detach().double().cpu() for n, p_ in layer.named_parameters()}
    # the oracle on the same values
    pr = {}
    for k, v in store.state_dict().items():
        v = v.double()
        if v.dim() == 2:
            v = v.to(torch.bfloat16).double()          # the matrices the kernels multiply with
        pr[k] = v.requires_grad_(True)
    xr = x.double().requires_grad_(True)
    yr = om.encoder_layer(pr, "enc", xr, None, {"num_heads": 4, "rate": 0.0}, training=True)
    yr.backward(dy.double())
    assert rel_l2(y.detach().double().cpu(), yr.detach()) < 1e-2
    assert rel_l2(xg.grad.double().cpu(), xr.grad) < 1e-2
    checked = 0
    for n_, gv in got.items():
        key = "enc." + n_.replace("ffn0", "ffn.0").replace("ffn1", "ffn.1")
        want = pr[key].grad
        if key.endswith("wk.bias"):
            assert float(gv.abs().max()) < 1e-2 * float(dy.double().abs().max())      # zero in exact arithmetic
            continue
        # the tail's own tensors (output projection, feed-forward block, LayerNorms) to 1e-2; the
        # gradients of the q / k / v projections also pass through the bf16 attention backward kernels
        # (dq, dk, dv rounded to bf16 before the weight-gradient product): measured 1.4e-2
        bound = 2e-2 if any(t in key for t in (".wq.", ".wk.", ".wv.")) else 1e-2
        assert rel_l2(gv, want) < bound, (key, rel_l2(gv, want))
        checked += 1
    assert checked >= 15


def test_layer_tail_repeats_bit_for_bit(ops, monkeypatch, tail_rows):
    """Twelve repetitions of the same layer on the same inputs give the same bits in every output and
    gradient: nothing in the tails (or in the weight-gradient table behind them) depends on timing.
    (Write-through stores issued as inline assembly once stored garbage here now and then: the wait
    states between a 16-byte store and a write to its data registers are the compiler's to insert.)"""
    first = None
    for rep in range(12):
        got = _encoder_layer_run(monkeypatch, True, 1024, 4, 64, 0.1, seed=31)
        junk = torch.randn(1 << 21, device=dev())          # other traffic between the repetitions
        del junk
        if first is None:
            first = got
            continue
        assert torch.equal(got[0], first[0]) and torch.equal(got[1], first[1]), rep
        for name in first[2]:
            assert torch.equal(got[2][name], first[2][name]), (rep, name)


def _encoder_stack_run(monkeypatch, fold, dff, B, T, rate, nlayers=3, seed=9):
    """Forward + backward pass of a bf16 encoder stack (d_model 256); ``fold``: the tail of a layer
    runs the q / k / v projections of the layer behind it."""
    from athena_amd.layers.commons import bind_all
    from athena_amd.layers.transformer import TransformerEncoder, TransformerEncoderLayer
    from athena_amd.runtime import ParamStore
    monkeypatch.setenv("ATHENA_LAYER_TAIL", "1")
    monkeypatch.setenv("ATHENA_TAIL_QKV", "1" if fold else "0")
    store = ParamStore(torch.bfloat16, device=dev())
    enc = TransformerEncoder([TransformerEncoderLayer(store, "enc.%d" % i, 256, 4, dff, rate) for i in range(nlayers)])
    store.finalize(seed=seed)
    bind_all(enc)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p_ in enc.named_parameters():
            if p_.dim() == 1:
                p_.add_(0.3 * torch.randn(p_.shape, generator=g).to(p_.device))
    store.refresh_shadows(force=True)
    x = torch.randn(B, T, 256, generator=g).to(dev()).to(torch.bfloat16).requires_grad_(True)
    dy = torch.randn(B, T, 256, generator=g).to(dev()).to(torch.bfloat16)
    store.grads.zero_()
    y = enc(x, None, training=True)
    y.backward(dy)
    torch.cuda.synchronize()
    grads = {n: p_.grad.detach().float().clone() for n, p_ in enc.named_parameters()}
    return y.detach().float(), x.grad.detach().float(), grads


@pytest.mark.parametrize("dff,B,T,rate", [(1024, 4, 64, 0.0), (512, 3, 50, 0.1), (768, 1, 7, 0.1)])
def test_layer_tail_runs_the_next_layers_projections(ops, monkeypatch, tail_rows, dff, B, T, rate):
    """A stack whose layer tails also run the q / k / v projections of the layer behind them
    (MultiHeadAttention.call, reference athena/layers/attention.py:150-153) against the same stack
    with the projections as a launch of their own: the same products on the same bf16 inputs, summed
    in another order."""
    y0, dx0, g0 = _encoder_stack_run(monkeypatch, False, dff, B, T, rate)
    y1, dx1, g1 = _encoder_stack_run(monkeypatch, True, dff, B, T, rate)
    assert torch.isfinite(y1).all() and torch.isfinite(dx1).all()
    assert rel_l2(y1, y0) < 4e-3, ("y", rel_l2(y1, y0))
    assert rel_l2(dx1, dx0) < 1e-2, ("dx", rel_l2(dx1, dx0))
    for name in g0:
        if name.endswith("wk.bias"):
            continue
        assert rel_l2(g1[name], g0[name]) < 1e-2, (name, rel_l2(g1[name], g0[name]))


def test_layer_tail_projection_values(ops):
    """The three extra products of ``layer_tail_fwd`` against ``y2 @ W + b`` computed from the y2 the
    same launch wrote."""
    from athena_amd.layers.commons import bind_all
    from athena_amd.layers.transformer import TransformerEncoderLayer
    from athena_amd.runtime import ParamStore
    from athena_amd.layers.fused import LN_EPS
    store = ParamStore(torch.bfloat16, device=dev())
    a = TransformerEncoderLayer(store, "a", 256, 4, 512, 0.0)
    b = TransformerEncoderLayer(store, "b", 256, 4, 512, 0.0)
    store.finalize(seed=4)
    bind_all(a)
    bind_all(b)
    g = torch.Generator().manual_seed(4)
    with torch.no_grad():
        for m in (b.self_attn.wq, b.self_attn.wk, b.self_attn.wv):
            m.bias.add_(torch.randn(256, generator=g).to(dev()))
    store.refresh_shadows(force=True)
    rows = 75
    c = torch.randn(rows, 256, generator=g).to(dev()).to(torch.bfloat16)
    x = torch.randn(rows, 256, generator=g).to(dev()).to(torch.bfloat16)
    nq = (b.self_attn.wq, b.self_attn.wk, b.self_attn.wv)
    t = ops.layer_tail_fwd(c, x, a.self_attn.dense, a.norm1, a.ffn0, a.ffn1, a.norm2, LN_EPS, 0.0, store.rng_state,
                           1, 2, 3, next_qkv=nq)
    torch.cuda.synchronize()
    y2 = t["y2"].double()
    for name, m in zip("qkv", nq):
        want = y2 @ m.kernel.detach().to(torch.bfloat16).double() + m.bias.detach().double()
        assert_close(t[name], want.cpu(), 6e-3, "tail projection " + name)


def _decoder_stack_run(monkeypatch, tails, fold, dff, B, L, T, rate, nlayers=3, seed=11):
    """Forward + backward pass of a bf16 decoder stack (d_model 256) attending to a memory; ``tails``:
    the one-launch tails, ``fold``: the projections / the last backward product of a layer's
    self-attention run by the tails of the layer in front."""
    from athena_amd.layers.attention import AttnMask
    from athena_amd.layers.commons import bind_all
    from athena_amd.layers.transformer import TransformerDecoder, TransformerDecoderLayer
    from athena_amd.runtime import ParamStore
    monkeypatch.setenv("ATHENA_LAYER_TAIL", "1")
    monkeypatch.setenv("ATHENA_LAYER_TAIL_DECODER", "1" if tails else "0")
    monkeypatch.setenv("ATHENA_TAIL_QKV", "1" if fold else "0")
    store = ParamStore(torch.bfloat16, device=dev())
    dec = TransformerDecoder([TransformerDecoderLayer(store, "dec.%d" % i, 256, 4, dff, rate) for i in range(nlayers)])
    store.finalize(seed=seed)
    bind_all(dec)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p_ in dec.named_parameters():
            if p_.dim() == 1:
                p_.add_(0.3 * torch.randn(p_.shape, generator=g).to(p_.device))
    store.refresh_shadows(force=True)
    tgt = torch.randn(B, L, 256, generator=g).to(dev()).to(torch.bfloat16).requires_grad_(True)
    mem = torch.randn(B, T, 256, generator=g).to(dev()).to(torch.bfloat16).requires_grad_(True)
    dy = torch.randn(B, L, 256, generator=g).to(dev()).to(torch.bfloat16)
    store.grads.zero_()
    y = dec(tgt, mem, tgt_mask=AttnMask(None, True), memory_mask=None, training=True)
    y.backward(dy)
    torch.cuda.synchronize()
    grads = {n: p_.grad.detach().float().clone() for n, p_ in dec.named_parameters()}
    return y.detach().float(), tgt.grad.detach().float(), mem.grad.detach().float(), grads


@pytest.mark.parametrize("dff,B,L,T,rate", [(1024, 4, 33, 64, 0.0), (512, 3, 20, 50, 0.1)])
def test_decoder_tails_and_handed_over_products(ops, monkeypatch, tail_rows, dff, B, L, T, rate):
    """A decoder stack three ways: separate Dense / LayerNorm launches; one-launch tails; tails that
    also run the self-attention projections of the layer behind them and, backwards, add the input
    gradient of those projections (reference athena/layers/transformer.py:374-388).  Same dropout
    decisions everywhere: the runs differ by bf16 rounding of intermediates."""
    sep = _decoder_stack_run(monkeypatch, False, False, dff, B, L, T, rate)
    tails = _decoder_stack_run(monkeypatch, True, False, dff, B, L, T, rate)
    fold = _decoder_stack_run(monkeypatch, True, True, dff, B, L, T, rate)
    # (what, candidate, reference, bound on the outputs, bound on the parameter gradients): the tails keep
    # intermediates in fp32 that the separate launches round to bf16, and three layers of random weights
    # amplify that; handing products over changes the order of a few fp32 additions only
    for what, got, ref, tol, tol_g in (("tails", tails, sep, 2e-2, 6e-2), ("fold", fold, tails, 6e-3, 1.2e-2)):
        for name, a, b in (("y", got[0], ref[0]), ("dtgt", got[1], ref[1]), ("dmem", got[2], ref[2])):
            assert torch.isfinite(a).all(), name
            assert rel_l2(a, b) < tol, (what, name, rel_l2(a, b))
        for name in ref[3]:
            if name.endswith("wk.bias"):
                continue
            assert rel_l2(got[3][name], ref[3][name]) < tol_g, (what, name, rel_l2(got[3][name], ref[3][name]))


def test_refresh_shadows(ops):
    from athena_amd.runtime import ParamStore, glorot_uniform
    st = ParamStore(torch.bfloat16, device=dev())
    st.register("a", (40, 72), glorot_uniform(40, 72), matrix=(40, 72))
    st.register("b", (7,), glorot_uniform(7, 7))
    st.register("c", (3, 3, 8, 13), glorot_uniform(72, 13), matrix=(72, 13))
    st.finalize(seed=3)
    st.refresh_shadows()
    for name in ("a", "c"):
        r, c = st.by_name[name].matrix
        w = st.view(name).reshape(r, c)
        s, s_t = st.shadow_views(name)
        assert torch.equal(s[:, :c], w.to(torch.bfloat16))
        assert torch.equal(s_t[:, :r], w.t().to(torch.bfloat16))
        assert float(s[:, c:].abs().sum()) == 0.0 and float(s_t[:, r:].abs().sum()) == 0.0


# ------------------------------------------------------------------------------------------ label preparation
def test_prepare_labels_matches_reference_helpers(ops):
    """One-launch label / mask preparation against the step-by-step helpers that restate
    athena/utils/misc.py:32-90 and speech_transformer.py:156-162."""
    from athena_amd.utils import misc
    g = torch.Generator().manual_seed(5)
    B, U, V, T = 7, 9, 50, 61
    labels = torch.randint(1, V - 1, (B, U), generator=g, dtype=torch.int32)
    lens = torch.tensor([9, 0, 3, 9, 1, 5, 8], dtype=torch.int32)
    for b in range(B):
        labels[b, lens[b]:] = 0
    labels[2, 1] = 0                                  # label id 0 inside a sequence
    in_len = torch.tensor([61 * 4, 1, 2, 3, 4, 5, 243], dtype=torch.int32)
    Tp = 61
    misc.StepLabels.last = None
    y0 = misc.insert_sos_in_labels(labels, V - 1)
    tgt = misc.insert_eos_in_labels(labels, V - 1, lens)
    tl = (((in_len + 1) // 2) + 1) // 2
    xm = (torch.arange(Tp)[None, :] >= tl[:, None]).to(torch.uint8)
    got = ops.prepare_labels(labels.to(dev()), lens.to(dev()), in_len.to(dev()), Tp, V - 1, V - 1)
    assert torch.equal(got[0].cpu(), y0)
    assert torch.equal(got[1].cpu(), tgt)
    assert torch.equal(got[2].cpu(), (y0 == 0).to(torch.uint8))
    assert torch.equal(got[3].cpu(), tl)
    assert torch.equal(got[4].cpu(), xm)


def test_copy_many_device_and_pinned(ops):
    """Several copies in one launch: device to device (aligned and not), device to pinned host
    memory, more pairs than one launch takes, an empty tensor."""
    g = torch.Generator().manual_seed(5)
    srcs = [torch.randn(n, generator=g).to(dev()) for n in (1, 7, 1024, 100003, 33, 5, 64, 129, 4097, 2)]
    srcs.append(torch.randint(0, 100, (32, 17), generator=g, dtype=torch.int32).to(dev()))
    srcs.append(torch.empty(0, device=dev()))
    odd = torch.randn(1000, generator=g).to(dev())[3:900]          # not 16-byte aligned
    srcs.append(odd.contiguous()[1:])
    dsts = [torch.zeros_like(s) for s in srcs]
    pinned = torch.zeros(4097, pin_memory=True)
    dsts[8] = pinned
    ops.copy_many(list(zip(srcs, dsts)))
    torch.cuda.synchronize()
    for s, d in zip(srcs, dsts):
        assert torch.equal(s.cpu(), d.cpu())
    with pytest.raises(RuntimeError):
        ops.copy_many([(srcs[0], torch.zeros(2, device=dev()))])


@pytest.mark.parametrize("M,N,K", [(1056, 256, 4240), (8000, 256, 4240), (330, 72, 2056), (64, 40, 512)])
def test_gemm_bf16_output_with_split_contraction(ops, M, N, K):
    """bf16 operands AND a bf16 output with the contraction left to split (split_k = 0): the partial
    sums go through the fp32 workspace and the second pass rounds once, so the result equals the
    unsplit product up to the order of the fp32 additions."""
    g = torch.Generator().manual_seed(31)
    a = to_cd(torch.randn(M, K, generator=g, dtype=torch.float64), torch.bfloat16)
    b = to_cd(torch.randn(N, K, generator=g, dtype=torch.float64) * 0.1, torch.bfloat16)
    ref = a.double().cpu() @ b.double().cpu().t()
    c0 = torch.empty(M, N, dtype=torch.bfloat16, device=dev())
    c1 = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=dev())
    ops.gemm(a, b, c0, M, N, K, trans_b=True)
    ops.gemm(a, b, c1, M, N, K, trans_b=True, split_k=0)
    torch.cuda.synchronize()
    assert_close(c1, ref, 6e-3, "bf16 output, split contraction")        # one bf16 rounding of values up to ~ max|ref|
    assert float((c1.double() - c0.double()).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
    # an epilogue cannot be split
    with pytest.raises(RuntimeError):
        ops.gemm(a, b, c1, M, N, K, trans_b=True, split_k=0, bias=torch.zeros(N, device=dev()))


# ------------------------------------------------------------------------------------------ accuracy scoring
def _ref_counts(ids, lens, labels, mode, special):
    """Per-utterance (errors, labels) by the oracle's loops."""
    out = []
    for b in range(ids.shape[0]):
        row = ids[b].tolist()
        if mode == 0:
            hyp, prev = [], None
            for t in range(int(lens[b])):
                c = row[t]
                if c != prev and c != special:
                    hyp.append(c)
                prev = c
        else:
            hyp = []
            for c in row:
                if special != 0 and c == special:
                    break
                hyp.append(c)
            hyp = [c for c in hyp if c != 0]
        ref = [int(x) for x in labels[b].tolist() if int(x) != 0]
        out.append((om.edit_distance(hyp, ref), len(ref)))
    return out


@pytest.mark.parametrize("B,L,U,V", [(5, 40, 7, 6), (32, 250, 32, 4233), (8, 750, 150, 50), (3, 90, 64, 9),
                                     (4, 300, 200, 5), (2, 17, 1, 4), (2, 255, 300, 7), (2, 256, 30, 7), (1, 1500, 1200, 11),
                                     (1, 3000, 40, 5)])
@pytest.mark.parametrize("mode", [0, 1])
def test_seq_errors_matches_oracle_loops(ops, B, L, U, V, mode):
    """athena_seq_errors against the oracle's greedy decode / validate_seqs + edit distance: ragged
    lengths, empty hypotheses, empty references, zeros inside the labels, one wave and several."""
    g = torch.Generator().manual_seed(B * 1000 + L + U + mode)
    special = V - 1
    if mode == 0:
        # runs of repeated ids so the collapse has something to merge; a quarter of the frames blank
        ids = torch.randint(0, V, (B, L), generator=g, dtype=torch.int32)
        rep = torch.rand(B, L, generator=g) < 0.4
        for t in range(1, L):
            ids[:, t] = torch.where(rep[:, t], ids[:, t - 1], ids[:, t])
        ids[torch.rand(B, L, generator=g) < 0.25] = special
    else:
        ids = torch.randint(0, V - 1, (B, L), generator=g, dtype=torch.int32)
        for b in range(B):                              # eos somewhere (or nowhere, or first)
            pos = int(torch.randint(0, 2 * L, (1,), generator=g))
            if pos < L:
                ids[b, pos:] = torch.randint(0, V, (L - pos,), generator=g, dtype=torch.int32)
                ids[b, pos] = special
        ids[0, 0] = special                             # empty hypothesis
    lens = torch.randint(0, L + 1, (B,), generator=g, dtype=torch.int32)
    lens[0] = L
    if B > 1:
        lens[1] = 0                                     # nothing decoded
    labels = torch.randint(0, max(V - 1, 2), (B, U), generator=g, dtype=torch.int32)   # zeros inside
    lab_len = torch.randint(0, U + 1, (B,), generator=g)
    for b in range(B):
        labels[b, int(lab_len[b]):] = 0
    if B > 2:
        labels[2] = 0                                   # empty reference
    want = _ref_counts(ids, lens, labels, mode, special)
    counts = torch.tensor([5, 7], dtype=torch.int64, device=dev())
    per = torch.full((B, 2), -1, dtype=torch.int32, device=dev())
    ops.seq_errors(ids.to(dev()), lens.to(dev()) if mode == 0 else None, labels.to(dev()), mode, special,
                   counts=counts, per_utt=per)
    assert per.cpu().tolist() == [list(w) for w in want]
    assert counts.cpu().tolist() == [5 + sum(w[0] for w in want), 7 + sum(w[1] for w in want)]


def test_seq_errors_eos_zero_and_limits(ops):
    ids = torch.tensor([[3, 0, 2, 0, 1]], dtype=torch.int32, device=dev())
    labels = torch.tensor([[3, 2, 2, 0]], dtype=torch.int32, device=dev())
    per = torch.zeros(1, 2, dtype=torch.int32, device=dev())
    ops.seq_errors(ids, None, labels, 1, 0, per_utt=per)          # eos == 0: only the zeros go
    assert per.cpu().tolist() == [[1, 3]]
    with pytest.raises(RuntimeError):                               # more ids per utterance than a wave's registers hold
        ops.seq_errors(torch.zeros(1, 4096, dtype=torch.int32, device=dev()), None, labels, 1, 0, per_utt=per)


def test_accuracy_metrics_score_on_the_device():
    """CTCAccuracy / Seq2SeqSparseCategoricalAccuracy fed with logits: the counts equal the oracle's
    (metrics.py:101-132), accumulate over updates, reset, and skip a batch whose logits are no
    longer than the labels + 1."""
    from athena_amd.metrics import CTCAccuracy, Seq2SeqSparseCategoricalAccuracy
    g = torch.Generator().manual_seed(11)
    B, T, V, U = 6, 30, 9, 5
    m, m2 = CTCAccuracy(), Seq2SeqSparseCategoricalAccuracy(eos=V - 1)
    err = tot = err2 = tot2 = 0.0
    for _ in range(3):
        logits = torch.randn(B, T, V, generator=g)
        att = torch.randn(B, U + 1, V, generator=g)
        labels = torch.randint(1, V - 1, (B, U), generator=g, dtype=torch.int32)
        lens = torch.randint(1, U + 1, (B,), generator=g, dtype=torch.int32)
        for b in range(B):
            labels[b, int(lens[b]):] = 0
        tl = torch.randint(U + 2, T + 1, (B,), generator=g, dtype=torch.int32)
        samples = {"output": labels, "output_length": lens}
        e, t = om.ctc_accuracy_counts(logits, samples, tl)
        e2, t2 = om.seq2seq_accuracy_counts(att, samples, V - 1)
        err, tot, err2, tot2 = err + e, tot + t, err2 + e2, tot2 + t2
        ds = {k: v.to(dev()) for k, v in samples.items()}
        m(logits.to(dev()), ds, tl.to(dev()))
        m2(att.to(dev()), ds)
    assert abs(m.result() - (1 - err / tot)) < 1e-12
    assert abs(m2.result() - (1 - err2 / tot2)) < 1e-12
    m.reset_states()
    assert m.result() != m.result()
    short = torch.randn(B, U + 1, V).to(dev())
    m(short, ds, tl.to(dev()))                                      # skipped (metrics.py:124-125)
    assert m.result() != m.result()


# ------------------------------------------------------------------------------------------ ffn activations, attention weights
def _oracle_params(store, cd):
    pr = {}
    for k, v in store.state_dict().items():
        v = v.double()
        if v.dim() == 2 and cd == torch.bfloat16:
            v = v.to(torch.bfloat16).double()
        pr[k] = v.requires_grad_(True)
    return pr


@pytest.mark.parametrize("activation", ["relu", "relu6", "elu", "selu", "leaky_relu", "sigmoid", "softplus", "softsign",
                                        "tanh", "gelu"])
def test_encoder_layer_ffn_activations(activation):
    """TransformerEncoderLayer with every feed-forward activation of the reference's table
    (athena/layers/commons.py:494-506, used at transformer.py:262-270) against the oracle layer in fp64:
    output 2e-5, every gradient 1e-4 (fp32 mode)."""
    from athena_amd.layers.commons import bind_all
    from athena_amd.layers.transformer import TransformerEncoderLayer
    from athena_amd.runtime import ParamStore
    d, h, dff, B, T = 64, 4, 80, 3, 11
    store = ParamStore(torch.float32, device=dev())
    layer = TransformerEncoderLayer(store, "enc", d, h, dff, 0.0, activation=activation)
    store.finalize(seed=3)
    bind_all(layer)
    g = torch.Generator().manual_seed(3)
    with torch.no_grad():
        for name, p_ in layer.named_parameters():
            p_.add_((0.3 if p_.dim() == 1 else 0.15) * torch.randn(p_.shape, generator=g).to(p_.device))
    store.refresh_shadows(force=True)
    x = torch.randn(B, T, d, generator=g) * 1.5
    dy = torch.randn(B, T, d, generator=g)
    xg = x.to(dev()).requires_grad_(True)
    store.grads.zero_()
    y = layer(xg, None, training=True)
    y.backward(dy.to(dev()))
    torch.cuda.synchronize()
    pr = _oracle_params(store, torch.float32)
    xr = x.double().requires_grad_(True)
    yr = om.encoder_layer(pr, "enc", xr, None, {"num_heads": h, "rate": 0.0, "ffn_activation": activation}, training=True)
    yr.backward(dy.double())
    assert_close(y, yr.detach(), 2e-5, "layer output (%s)" % activation)
    assert_close(xg.grad, xr.grad, 1e-4, "input gradient (%s)" % activation)
    for n_, p_ in layer.named_parameters():
        key = "enc." + n_.replace("ffn0", "ffn.0").replace("ffn1", "ffn.1")
        if key.endswith("wk.bias"):
            continue
        assert_close(p_.grad, pr[key].grad, 1e-4, key)


def test_unknown_ffn_activation_is_refused():
    from athena_amd.layers.transformer import TransformerEncoderLayer
    from athena_amd.runtime import ParamStore
    with pytest.raises(NotImplementedError):
        TransformerEncoderLayer(ParamStore(torch.float32, device=dev()), "enc", 32, 4, 64, 0.0, activation="glu")


@pytest.mark.parametrize("cd", DTYPES)
def test_attention_weights_on_request(cd):
    """MultiHeadAttention(..., return_attention_weights=True): the [B, h, Tq, Tk] weights the reference
    returns beside the output (athena/layers/attention.py:161,171) against the oracle's, with a key
    padding mask, with the causal mask, and for a unidirectional layer with look-ahead; and the
    decoder stack's list of cross-attention weights (transformer.py:225-231)."""
    from athena_amd.layers.attention import AttnMask, MultiHeadAttention
    from athena_amd.layers.commons import bind_all
    from athena_amd.layers.transformer import TransformerDecoder, TransformerDecoderLayer
    from athena_amd.runtime import ParamStore
    d, h, B, Tq, Tk = 64, 4, 2, 9, 13
    store = ParamStore(cd, device=dev())
    mha = MultiHeadAttention(store, "a", d, h)
    uni = MultiHeadAttention(store, "u", d, h, unidirectional=True, look_ahead=2)
    dec = TransformerDecoder([TransformerDecoderLayer(store, "dec.%d" % i, d, h, 96, 0.0) for i in range(2)])
    store.finalize(seed=6)
    for m in (mha, uni, dec):
        bind_all(m)
    g = torch.Generator().manual_seed(6)
    with torch.no_grad():
        for m in (mha, uni, dec):
            for p_ in m.parameters():
                p_.add_((0.3 if p_.dim() == 1 else 0.2) * torch.randn(p_.shape, generator=g).to(p_.device))
    store.refresh_shadows(force=True)
    pr = _oracle_params(store, cd)
    tol = 1e-5 if cd == torch.float32 else 2e-2
    q = rounded(torch.randn(B, Tq, d, generator=g, dtype=torch.float64), cd)
    kv = rounded(torch.randn(B, Tk, d, generator=g, dtype=torch.float64), cd)
    pad = torch.zeros(B, Tk, dtype=torch.uint8)
    pad[1, 9:] = 1
    fmask = pad.double()[:, None, None, :]
    with torch.no_grad():
        kvd, qd = to_cd(kv, cd), to_cd(q, cd)
        out, w = mha(kvd, kvd, qd, AttnMask(pad.to(dev()), False), return_attention_weights=True)
        out_r, w_r = om.multi_head_attention(pr, "a", kv, kv, q, fmask, h)
        assert tuple(w.shape) == (B, h, Tq, Tk)
        assert_close(w, w_r, tol, "weights, padding mask")
        assert_close(out, out_r, TOL[cd], "output unchanged by the request")
        assert mha(kvd, kvd, qd, None)[1] is None
        # causal self-attention
        causal = torch.triu(torch.ones(Tq, Tq, dtype=torch.float64), diagonal=1)[None, None]
        _, w = mha(qd, qd, qd, AttnMask(None, True), return_attention_weights=True)
        _, w_r = om.multi_head_attention(pr, "a", q, q, q, causal, h)
        assert_close(w, w_r, tol, "weights, causal mask")
        # unidirectional layer, look-ahead 2
        _, w = uni(qd, qd, qd, None, return_attention_weights=True)
        _, w_r = om.multi_head_attention(pr, "u", q, q, q, None, h, unidirectional=True, look_ahead=2)
        assert_close(w, w_r, tol, "weights, band")
        # decoder stack
        tgt = qd
        y, ws = dec(tgt, kvd, tgt_mask=AttnMask(None, True), memory_mask=AttnMask(pad.to(dev()), False),
                    return_attention_weights=True, training=False)
        y_plain = dec(tgt, kvd, tgt_mask=AttnMask(None, True), memory_mask=AttnMask(pad.to(dev()), False),
                      training=False)
        assert len(ws) == 2 and all(tuple(w_.shape) == (B, h, Tq, Tk) for w_ in ws)
        assert torch.equal(y, y_plain)
        t_r = q
        for i in range(2):
            t_next, w_r = om.decoder_layer(pr, "dec.%d" % i, t_r, kv, causal, fmask, {"num_heads": h, "rate": 0.0}, False)
            assert_close(ws[i], w_r, tol if cd == torch.float32 else 4e-2, "cross-attention weights of layer %d" % i)
            t_r = t_next


@pytest.mark.parametrize("rows,nblock", [(1, 1), (37, 3), (1056, 12), (8005, 12), (2049, 20)])
def test_panel_gemm_against_fp64(ops, rows, nblock):
    """x[rows, 256] times a list of 256 x 256 kernels in one launch (``athena_panel_gemm``: the key / value
    Dense layers of every decoder layer's cross-attention on the encoder output, reference
    athena/layers/attention.py:150-152) against fp64 products of the bf16-rounded operands: 16- and 32-row
    workgroups, a last block of rows that is not full, one block without bias, outputs side by side in
    one buffer and an input that is a column slice of a wider tensor."""
    from athena_amd.layers.commons import Dense, bind_all
    from athena_amd.runtime import ParamStore
    store = ParamStore(torch.bfloat16, device=dev())

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = torch.nn.ModuleList([Dense(store, "d%d" % i, 256, 256, None, tiled=True) for i in range(nblock)])
    holder = Holder()
    store.finalize(seed=5)
    bind_all(holder)
    g = torch.Generator().manual_seed(rows)
    with torch.no_grad():
        for m in holder.layers:
            m.bias.copy_(torch.randn(256, generator=g).to(dev()))
    store.refresh_shadows(force=True)
    wide = torch.randn(rows, 320, generator=g).to(dev()).to(torch.bfloat16)
    x = wide[:, 64:]
    out = torch.full((rows, 256 * nblock + 8), 7.0, dtype=torch.bfloat16, device=dev())
    blocks = [(m.tiled_t, m.bias if i != 1 else None, out[:, 256 * i:], out.stride(0)) for i, m in enumerate(holder.layers)]
    ops.panel_gemm(x, rows, blocks)
    torch.cuda.synchronize()
    assert torch.all(out[:, 256 * nblock:] == 7.0)
    x64 = x.double()
    for i, m in enumerate(holder.layers):
        want = x64 @ m.kernel.detach().to(torch.bfloat16).double()
        if i != 1:
            want = want + m.bias.detach().double()
        got = out[:, 256 * i:256 * (i + 1)].double()
        assert (got - want).abs().max() <= 2 ** -8 * want.abs().max() + 1e-6, i
        assert rel_l2(got, want) < 4e-3, i


def test_panel_gemm_as_dense_input_gradient(ops):
    """dx[rows, 768] = dy[rows, 256] kernel^T of a Dense(768 -> 256) through ``athena_panel_gemm`` (blocks = the
    kernel's rows in groups of 256, tiled copy with the kernel's rows as rows) against ``athena_gemm``."""
    from athena_amd.layers.commons import Dense, bind_all
    from athena_amd.runtime import ParamStore
    store = ParamStore(torch.bfloat16, device=dev())
    dense = Dense(store, "x", 768, 256, None, tiled=True)
    store.finalize(seed=9)
    bind_all(dense)
    store.refresh_shadows(force=True)
    rows = 4000 + 3
    dy = torch.randn(rows, 256, generator=torch.Generator().manual_seed(2)).to(dev()).to(torch.bfloat16)
    dx = torch.empty(rows, 768, dtype=torch.bfloat16, device=dev())
    ops.panel_gemm(dy, rows, [(dense.tiled[65536 * c:], None, dx[:, 256 * c:], 768) for c in range(3)])
    want = dy.double() @ dense.kernel.detach().to(torch.bfloat16).double().t()
    assert (dx.double() - want).abs().max() <= 2 ** -8 * want.abs().max() + 1e-6
    assert rel_l2(dx.double(), want) < 4e-3


@pytest.mark.parametrize("rows,nchunk,short,act,accumulate", [(37, 2, 0, 0, False), (1056, 12, 0, 1, False),
                                                              (8005, 20, 0, 1, False), (2049, 3, 112, 0, True),
                                                              (300, 17, 112, 0, False)])
def test_panel_gemm_k_against_fp64(ops, rows, nchunk, short, act, accumulate):
    """out[rows, 256] (+)= act(a[rows, K] kernel + bias) with the output rows resident over the whole
    contraction (``athena_panel_gemm_k``: the x_net Dense, reference athena/models/speech_transformer.py:95)
    against fp64 on the bf16-rounded operands: 16- and 32-row workgroups, relu6, accumulation onto a bf16
    buffer, and an ``a`` whose last 256-column piece is short (the kernel rows it lacks are zero)."""
    from athena_amd.layers.commons import Dense, bind_all
    from athena_amd.runtime import ParamStore
    from athena_amd._lib import ACT_NONE, ACT_RELU6
    K = 256 * nchunk
    store = ParamStore(torch.bfloat16, device=dev())
    dense = Dense(store, "x", K, 256, None, tiled=True)
    store.finalize(seed=4)
    bind_all(dense)
    g = torch.Generator().manual_seed(rows + nchunk)
    kcols = K - short
    with torch.no_grad():
        dense.kernel.mul_(8.0)
        dense.kernel[kcols:].zero_()
        dense.bias.copy_(torch.randn(256, generator=g).to(dev()))
    store.refresh_shadows(force=True)
    a = torch.randn(rows, kcols, generator=g).to(dev()).to(torch.bfloat16)
    base = torch.randn(rows, 256, generator=g).to(dev()).to(torch.bfloat16)
    out = base.clone() if accumulate else torch.empty(rows, 256, dtype=torch.bfloat16, device=dev())
    ops.panel_gemm_k(a, rows, kcols, [(dense.tiled_t[8 * c * 512:], K) for c in range(nchunk)], out, 256,
                     bias=dense.bias, act=ACT_RELU6 if act else ACT_NONE, accumulate=accumulate)
    want = a.double() @ dense.kernel.detach().to(torch.bfloat16).double()[:kcols] + dense.bias.detach().double()
    if act:
        want = want.clamp(0.0, 6.0)
    if accumulate:
        want = want + base.double()
    assert torch.isfinite(out).all()
    assert (out.double() - want).abs().max() <= 2 ** -7 * want.abs().max() + 1e-6
    assert rel_l2(out.double(), want) < 5e-3


def test_panel_gemm_k_with_a_matrix_per_piece(ops):
    """The sum over several Dense layers of dy_l kernel_l^T (the gradient of the encoder output through all
    decoder layers' key / value projections, backward of reference athena/layers/attention.py:151-152): one piece
    per 256 x 256 kernel, its tiled copy with the kernel's rows as rows."""
    from athena_amd.layers.commons import Dense, bind_all
    from athena_amd.runtime import ParamStore
    store = ParamStore(torch.bfloat16, device=dev())

    class Holder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.layers = torch.nn.ModuleList([Dense(store, "d%d" % i, 256, 256, None, tiled=True) for i in range(4)])
    holder = Holder()
    store.finalize(seed=6)
    bind_all(holder)
    store.refresh_shadows(force=True)
    rows = 4100
    dy = torch.randn(rows, 1024, generator=torch.Generator().manual_seed(3)).to(dev()).to(torch.bfloat16)
    out = torch.empty(rows, 256, dtype=torch.bfloat16, device=dev())
    ops.panel_gemm_k(dy, rows, 1024, [(m.tiled, 256) for m in holder.layers], out, 256)
    want = sum(dy[:, 256 * i:256 * (i + 1)].double() @ m.kernel.detach().to(torch.bfloat16).double().t()
               for i, m in enumerate(holder.layers))
    assert (out.double() - want).abs().max() <= 2 ** -7 * want.abs().max() + 1e-6
    assert rel_l2(out.double(), want) < 5e-3
