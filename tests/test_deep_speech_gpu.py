"""The DeepSpeech sibling (reference athena/models/deep_speech.py:29-96) against the CPU oracle
(oracle/deep_speech.py): wide-kernel convolution kernels, BatchNormalization without relu6, the
recurrent layers, the whole training step and a few optimizer steps through the solver."""
import copy

import pytest
import torch
import torch.nn.functional as F

from oracle import deep_speech as ods
from oracle import model as om
from tests.modelutil import DataDescriptions
from tests.test_model_gpu import oracle_params
from tests.util import TOL, assert_close, dev, rel_err, rel_l2, rounded, to_cd

pytestmark = pytest.mark.gpu

CONV_CASES = [
    # B, H, W, Cin, Cout, KH, KW, sh, sw
    (2, 19, 10, 8, 16, 21, 11, 2, 1),      # the model's second convolution, small
    (2, 12, 7, 16, 8, 5, 3, 2, 2),
    (1, 30, 20, 64, 64, 21, 11, 2, 1),     # its shipped channel count (examples/asr/hkust/configs/deep_speech.json)
    (3, 9, 9, 8, 24, 3, 3, 1, 1),
]


@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv2d_any_kernel(case, cd):
    """Forward (with the BatchNorm column statistics), input gradient (with and without the relu6 gate)
    and weight gradient of the implicit-GEMM convolution against torch's conv2d in float64."""
    from athena_amd import ops
    B, H, W, Cin, Cout, KH, KW, sh, sw = case
    g = torch.Generator().manual_seed(7)
    x = rounded(torch.randn(B, H, W, Cin, generator=g, dtype=torch.float64), cd)
    w = rounded(torch.randn(KH, KW, Cin, Cout, generator=g, dtype=torch.float64) / (KH * KW * Cin) ** 0.5, cd)
    Ho, Wo = ops.conv2d_out_shape(H, W, KH, KW, sh, sw)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    ref = ods.conv2d_same(xr, wr, (sh, sw))
    assert tuple(ref.shape) == (B, Ho, Wo, Cout)
    dy = rounded(torch.randn(ref.shape, generator=g, dtype=torch.float64), cd)
    ref.backward(dy)
    tol = TOL[cd]
    out = torch.empty((B, Ho, Wo, Cout), dtype=cd, device=dev())
    stats = torch.empty(2 * Cout, dtype=torch.float64, device=dev())
    xd, wd = to_cd(x, cd), to_cd(w, cd)
    ops.conv2d_fwd(xd, wd, out, B, H, W, Cin, Cout, KH, KW, sh, sw, stats)
    assert_close(out, ref.detach(), tol, "conv2d fwd")
    o64 = out.double().cpu().reshape(-1, Cout)
    assert_close(stats[:Cout], o64.sum(0), 1e-5 if cd == torch.float32 else 2e-2, "column sums")
    assert_close(stats[Cout:], (o64 * o64).sum(0), 1e-5 if cd == torch.float32 else 2e-2, "column sums of squares")
    dyd = to_cd(dy, cd)
    dx = torch.empty_like(xd)
    ops.conv2d_dgrad(dyd, wd, dx, B, H, W, Cin, Cout, KH, KW, sh, sw)
    assert_close(dx, xr.grad, tol, "conv2d dgrad")
    gate = rounded(torch.rand(B, H, W, Cin, generator=g, dtype=torch.float64) * 8 - 1, cd).clamp(0, 6)
    ops.conv2d_dgrad(dyd, wd, dx, B, H, W, Cin, Cout, KH, KW, sh, sw, relu6_out=to_cd(gate, cd))
    assert_close(dx, xr.grad * ((gate > 0) & (gate < 6)), tol, "conv2d dgrad with relu6 gate")
    dw = torch.empty((KH, KW, Cin, Cout), dtype=torch.float32, device=dev())
    ops.conv2d_wgrad(xd, dyd, dw, B, H, W, Cin, Cout, KH, KW, sh, sw)
    assert_close(dw, wr.grad, 1e-5 if cd == torch.float32 else 1e-2, "conv2d wgrad")
    assert rel_l2(dw, wr.grad) <= (1e-5 if cd == torch.float32 else 1e-2)


@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_im2col_one_channel(cd):
    """The explicit im2col of the one-channel first layer: multiplying it with the kernel as a matrix
    reproduces the convolution; padding columns are zero."""
    from athena_amd import ops
    B, H, W, KH, KW = 2, 23, 11, 41, 11
    g = torch.Generator().manual_seed(8)
    x = torch.randn(B, H, W, 1, generator=g)
    w = torch.randn(KH, KW, 1, 4, generator=g, dtype=torch.float64)
    Ho, Wo = ops.conv2d_out_shape(H, W, KH, KW, 2, 2)
    K, ld = KH * KW, 456
    col = torch.full((B * Ho * Wo, ld), 7.0, dtype=cd, device=dev())
    ops.im2col(x.to(dev()), col, ld, B, H, W, 1, KH, KW, 2, 2)
    c = col.double().cpu()
    assert float(c[:, K:].abs().max()) == 0.0
    ref = ods.conv2d_same(rounded(x.double(), cd), w, (2, 2)).reshape(-1, 4)
    assert_close(c[:, :K] @ w.reshape(K, 4), ref, 1e-12, "im2col product")


@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_batchnorm_without_relu6(cd):
    from athena_amd import ops
    rows, C = 777, 24
    g = torch.Generator().manual_seed(9)
    x = rounded(torch.randn(rows, C, generator=g, dtype=torch.float64) * 2 + 0.5, cd)
    gamma = torch.rand(C, generator=g, dtype=torch.float64) + 0.5
    beta = torch.randn(C, generator=g, dtype=torch.float64)
    dy = rounded(torch.randn(rows, C, generator=g, dtype=torch.float64), cd)
    xr, gr, br = x.clone().requires_grad_(True), gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref, mean, var = om.batch_norm_train(xr.view(1, 1, rows, C), gr, br)
    ref = ref.view(rows, C)
    ref.backward(dy)
    xd = to_cd(x, cd)
    stats = torch.empty(2 * C, dtype=torch.float64, device=dev())
    ops.bn_colstats(xd, rows, C, stats)
    assert_close(stats[:C], x.sum(0), 1e-12, "sums")
    assert_close(stats[C:], (x * x).sum(0), 1e-12, "sums of squares")
    mr = torch.empty(2 * C, dtype=torch.float32, device=dev())
    mm, mv = torch.zeros(C, device=dev()), torch.ones(C, device=dev())
    ops.bn_finalize(stats, rows, C, 1e-3, 0.99, mr, mm, mv)
    assert_close(mm, 0.01 * mean.detach(), 1e-5, "moving mean")
    gd, bd = gamma.float().to(dev()), beta.float().to(dev())
    y = torch.empty_like(xd)
    ops.bn_apply(xd, mr, gd, bd, rows, C, y)
    assert_close(y, ref.detach(), TOL[cd], "bn apply")
    assert float(y.min()) < 0.0                      # no relu6 in this variant
    yi = torch.empty_like(xd)
    ops.bn_infer(xd, mm, mv, gd, bd, 1e-3, rows, C, yi)
    want = om.batch_norm_infer(x, gamma, beta, mm.double().cpu(), mv.double().cpu())
    assert_close(yi, want, TOL[cd], "bn infer")
    dx = torch.empty_like(xd)
    dg, db = torch.empty(C, device=dev()), torch.empty(C, device=dev())
    ops.bn_bwd(to_cd(dy, cd), xd, mr, gd, bd, rows, C, ops.bn_relu6_bwd_scratch(rows, C, dev()), dx, dg, db)
    assert_close(dx, xr.grad, TOL[cd], "bn dx")
    assert_close(dg, gr.grad, 1e-4, "bn dgamma")
    assert_close(db, br.grad, 1e-4, "bn dbeta")


@pytest.mark.parametrize("cd", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("cell", ["gru", "lstm"])
@pytest.mark.parametrize("shape", [(3, 9, 16, 24), (2, 33, 40, 64), (37, 5, 24, 1680)],
                         ids=["short", "longer", "two-row-blocks-wide"])
def test_recurrent_layer_kernels(cell, cd, shape):
    """athena_rnn_fwd / athena_rnn_bwd (the serial part) with the projections done as plain matrix
    products here, against the oracle's layer under autograd: outputs, input gradient and the three
    weight gradients."""
    from athena_amd import ops
    B, T, K, H = shape
    G = 4 if cell == "lstm" else 3
    code = ops.RNN_LSTM if cell == "lstm" else ops.RNN_GRU
    g = torch.Generator().manual_seed(11)
    x = rounded(torch.randn(B, T, K, generator=g, dtype=torch.float64), cd)
    W = rounded(torch.randn(K, G * H, generator=g, dtype=torch.float64) / K ** 0.5, cd)
    U = rounded(torch.randn(H, G * H, generator=g, dtype=torch.float64) / H ** 0.5, cd)
    bias = torch.randn((G * H,) if cell == "lstm" else (2, G * H), generator=g, dtype=torch.float64) * 0.2
    dy = rounded(torch.randn(B, T, H, generator=g, dtype=torch.float64), cd)
    leaves = [t.clone().requires_grad_(True) for t in (x, W, U, bias)]
    ref = (ods.lstm_layer if cell == "lstm" else ods.gru_layer)(*leaves)
    ref.backward(dy)
    tol = 3e-5 if cd == torch.float32 else 3e-2
    b_in = bias if cell == "lstm" else bias[0]
    xw = to_cd(x.reshape(B * T, K) @ W + b_in, cd)          # the library is handed the projection
    xw_seen = xw.double().cpu().view(B, T, G * H)
    ud = to_cd(U, cd)
    b_rec = None if cell == "lstm" else bias[1].float().to(dev())
    hs = torch.full((B, T + 1, H), 3.0, dtype=cd, device=dev())
    cs = torch.full((B, T + 1, H), 3.0, dtype=torch.float32, device=dev()) if cell == "lstm" else None
    rh = torch.empty((B, T, H), dtype=cd, device=dev()) if cell == "gru" else None
    y = torch.empty((B, T, H), dtype=cd, device=dev())
    gates = torch.empty((B, T, G * H), dtype=cd, device=dev())
    ws = ops.rnn_workspace(code, B, H, dev())
    ops.rnn_fwd(code, xw, ud, b_rec, hs, cs, y, gates, rh, B, T, H, ws)
    assert_close(y, ref.detach(), tol, "rnn outputs")
    if cd == torch.bfloat16:
        # with the transposed copy of the recurrent kernel a step is one launch (product and gates together)
        y2, hs2, gates2 = torch.empty_like(y), torch.full_like(hs, 3.0), torch.empty_like(gates)
        cs2 = torch.full_like(cs, 3.0) if cs is not None else None
        rh2 = torch.empty_like(rh) if rh is not None else None
        ops.rnn_fwd(code, xw, ud, b_rec, hs2, cs2, y2, gates2, rh2, B, T, H, ws, u_t=ud.t().contiguous())
        assert_close(y2, ref.detach(), tol, "rnn outputs, one launch per step")
        assert_close(gates2, gates, 2e-2, "gates of the two forward paths")
        assert float(hs2[:, 0].abs().max()) == 0.0 and torch.equal(hs2[:, 1:], y2)
    assert float(hs[:, 0].abs().max()) == 0.0 and torch.equal(hs[:, 1:], y)
    dxw = torch.empty((B, T, G * H), dtype=cd, device=dev())
    dhu = torch.full((B, T + 1, G * H), 5.0, dtype=cd, device=dev())
    ops.rnn_bwd(code, to_cd(dy, cd), ud, hs, cs, gates, rh, dxw, dhu, B, T, H, ws)
    assert float(dhu[:, T].abs().max()) == 0.0
    dxw64, dhu64, hs64 = dxw.double().cpu(), dhu.double().cpu(), hs.double().cpu()
    gtol = 1e-4 if cd == torch.float32 else 5e-2
    assert_close(dxw64.reshape(B * T, -1) @ W.t(), leaves[0].grad.reshape(B * T, K), gtol, "dx")
    assert_close(x.reshape(B * T, K).t() @ dxw64.reshape(B * T, -1), leaves[1].grad, gtol, "dW")
    assert_close(hs64.reshape(-1, H).t() @ dhu64.reshape(-1, G * H), leaves[2].grad, gtol, "dU")
    if cell == "lstm":
        assert_close(dxw64.sum((0, 1)), leaves[3].grad, gtol, "dbias")
    else:
        assert_close(dxw64.sum((0, 1)), leaves[3].grad[0], gtol, "input bias gradient")
        assert_close(dhu64.sum((0, 1)), leaves[3].grad[1], gtol, "recurrent bias gradient")
    del xw_seen


DS_CFG = {"conv_filters": 8, "rnn_hidden_size": 32, "num_rnn_layers": 2, "rnn_type": "gru"}


def _model(precision, cfg=None, feat=12, vocab=19, seed=3):
    from athena_amd import DeepSpeechModel
    return DeepSpeechModel(DataDescriptions(vocab, feat), copy.deepcopy(cfg or DS_CFG), precision=precision, seed=seed)


@pytest.mark.parametrize("rnn_type", ["gru", "lstm", "cudnngru"])
def test_deep_speech_fp32_step_matches_oracle(rnn_type):
    """Logits, CTC loss, CTC accuracy and every gradient of one training-mode step (ragged lengths)."""
    cfg = dict(DS_CFG, rnn_type=rnn_type)
    model = _model("fp32", cfg)
    raw = om.make_samples(3, 46, 12, 20, 4, seed=5, full_length=False)
    s = model.prepare_samples(raw)
    p = {k: v.double() for k, v in model.store.state_dict().items()}
    s64 = {k: (v.double() if v.is_floating_point() else v) for k, v in raw.items()}
    loss_ref, logits_ref, grads = ods.loss_and_grads(p, s64, cfg)
    out = model(s, training=True)
    assert tuple(out.shape) == tuple(logits_ref.shape) == (3, 12, 20)
    assert_close(out, logits_ref, 5e-5, "logits")
    loss, metrics = model.get_loss(out, s, training=True)
    assert_close(loss, loss_ref, 1e-5, "ctc loss")
    assert "CTCAccuracy" in metrics and float(metrics["CTCAccuracy"]) == float(metrics["CTCAccuracy"])
    loss.backward()
    assert set(grads) == {sl.name for sl in model.store.train_slots}
    for name, gref in grads.items():
        got = model.store.grad_view(name)
        assert rel_err(got, gref) <= 1e-4, (name, rel_err(got, gref))
        assert rel_l2(got, gref) <= 1e-4, (name, rel_l2(got, gref))
    # inference mode runs on the moving statistics the training pass has just updated
    stats = {}
    ods.forward(p, s64, cfg, training=True, bn_stats_out=stats)
    p2 = dict(p)
    for name, (mean, var, n) in stats.items():
        p2[name + ".moving_mean"] = 0.01 * mean
        p2[name + ".moving_variance"] = 0.99 + 0.01 * var * n / (n - 1)
    assert_close(model(s, training=False), ods.forward(p2, s64, cfg, training=False), 5e-5, "inference logits")


def test_deep_speech_bf16_close_to_fp32_oracle_and_trains():
    """bf16 compute: logits within bf16 accuracy of the float64 oracle; a few optimizer steps through
    the solver (HIP-graph replay of the whole step, per-step recurrence included) lower the loss."""
    from athena_amd import BaseSolver, WarmUpAdam
    model = _model("bf16", seed=4)
    raw = om.make_samples(4, 60, 12, 20, 5, seed=6, full_length=False)
    s = model.prepare_samples(raw)
    p = {k: v.double() for k, v in model.store.state_dict().items()}
    s64 = {k: (v.double() if v.is_floating_point() else v) for k, v in raw.items()}
    ref = ods.forward(p, s64, DS_CFG, training=True)
    out = model(s, training=True)
    scale = float(ref.abs().max())
    assert float((out.double().cpu() - ref).pow(2).mean().sqrt()) <= 0.03 * scale
    solver = BaseSolver(model, WarmUpAdam({"d_model": 32, "warmup_steps": 20, "k": 0.3}),
                        config={"clip_norm": 100.0, "enable_tf_function": True, "log_interval": 10 ** 9})
    losses = [float(solver.train_step(s)[0]) for _ in range(8)]
    assert all(l == l and l > 0 for l in losses) and losses[-1] < losses[0], losses


def test_deep_speech_shipped_config_dimensions():
    """The layer sizes of the shipped recipe (examples/asr/hkust/configs/deep_speech.json: 64 filters,
    hidden 1680, GRU) on a short batch with 40-dim features: one fp32 training step against the oracle
    with two recurrent layers instead of six."""
    cfg = {"conv_filters": 64, "rnn_hidden_size": 1680, "num_rnn_layers": 2, "rnn_type": "cudnngru"}
    model = _model("fp32", cfg, feat=40, vocab=30, seed=6)
    raw = om.make_samples(2, 40, 40, 31, 3, seed=7, full_length=True)
    s = model.prepare_samples(raw)
    p = {k: v.double() for k, v in model.store.state_dict().items()}
    s64 = {k: (v.double() if v.is_floating_point() else v) for k, v in raw.items()}
    loss_ref, logits_ref, grads = ods.loss_and_grads(p, s64, cfg)
    out = model(s, training=True)
    assert_close(out, logits_ref, 1e-4, "logits")
    loss, _ = model.get_loss(out, s, training=True)
    assert_close(loss, loss_ref, 1e-5, "ctc loss")
    loss.backward()
    for name, gref in grads.items():
        assert rel_l2(model.store.grad_view(name), gref) <= 2e-4, (name, rel_l2(model.store.grad_view(name), gref))


__all__ = ["F"]
