"""Masked predictive coding (reference athena/models/masked_pc.py, MPCLoss of athena/loss.py:83-105)
against the CPU oracle: loss kernel, model step, weight hand-over to the speech transformer."""
import copy

import pytest
import torch

from oracle import model as om
from tests.modelutil import DataDescriptions
from tests.test_model_gpu import ZERO_GRAD, oracle_params
from tests.util import assert_close, dev, rel_err, rel_l2

pytestmark = pytest.mark.gpu

MPC_CFG = {"num_filters": 16, "d_model": 32, "num_heads": 4, "num_encoder_layers": 2, "dff": 64, "rate": 0.0,
           "chunk_size": 1, "keep_probability": 0.7}


def _mpc_model(precision, feat=12, cfg=None):
    from athena_amd import MaskedPredictCoding
    dd = DataDescriptions(feat, feat)          # an MPC dataset reports the feature size as num_class
    return MaskedPredictCoding(dd, copy.deepcopy(cfg or MPC_CFG), precision=precision, seed=3)


def _batch(B=3, T=50, feat=12, seed=4):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, feat, 1, generator=g)
    n = torch.tensor([T, T - 7, T - 19][:B], dtype=torch.int32)
    for b in range(B):
        x[b, n[b]:] = 0.0
    return {"input": x, "input_length": n, "output": torch.zeros(B, 1, dtype=torch.int32),
            "output_length": torch.ones(B, dtype=torch.int32)}


def test_mpc_loss_kernel_matches_oracle():
    """Loss value and gradient on seeded tensors, ragged lengths, rows padded to 8 columns."""
    from athena_amd.loss import MPCLoss
    g = torch.Generator().manual_seed(1)
    B, Tp, N = 3, 11, 44
    logits = torch.randn(B, Tp, N, generator=g)
    target = torch.randn(B, Tp, N, generator=g)
    masked = target * (torch.rand(B, Tp, 1, generator=g) < 0.6)
    length = torch.tensor([11, 7, 1], dtype=torch.int32)
    lr = logits.double().requires_grad_(True)
    ref = om.mpc_loss(lr, {"output": target.double(), "input": masked.double()}, length)
    ref.backward()
    ld = logits.to(dev()).requires_grad_(True)
    samples = {"output": target.to(dev()), "input": masked.to(dev())}
    loss = MPCLoss()(ld, samples, length.to(dev()))
    assert_close(loss, ref.detach(), 1e-6, "mpc loss")
    (loss * 2.5).backward()
    assert_close(ld.grad, 2.5 * lr.grad, 1e-6, "mpc loss gradient")


def test_mpc_prepare_samples_masks_whole_chunks():
    model = _mpc_model("fp32")
    raw = _batch(T=51)
    s = model.prepare_samples(raw)
    assert s["input"].shape == (3, 48, 12, 1) and s["output"].shape == (3, 48, 12)
    assert s["input_length"].tolist() == [48, 44, 32] and s["output_length"].tolist() == [48, 44, 32]
    assert torch.equal(s["output"].cpu().reshape(3, 48, 12, 1), raw["input"][:, :48])
    kept = (s["input"].cpu() == raw["input"][:, :48]).reshape(3, 12, 4 * 12).all(-1)       # chunk = 4 frames
    gone = (s["input"].cpu() == 0).reshape(3, 12, 4 * 12).all(-1)
    assert bool((kept | gone).all()) and bool(gone[0].any()) and bool(kept[0].any())


def test_mpc_fp32_step_matches_oracle():
    model = _mpc_model("fp32")
    s = model.prepare_samples(_batch())
    p = oracle_params(model)
    s64 = {k: (v.double().cpu() if v.is_floating_point() else v.cpu()) for k, v in s.items()}
    ocfg = dict(MPC_CFG, num_class=48, feat_dim=12)
    ref = om.mpc_forward(p, s64, ocfg, training=True)
    loss_ref = om.mpc_loss(ref, s64, om.compute_logit_length(s64["input_length"]))
    loss_ref.backward()
    out = model(s, training=True)
    assert tuple(out.shape) == (3, 12, 48)
    assert_close(out, ref.detach(), 2e-5, "mpc output")
    loss, metrics = model.get_loss(out, s, training=True)
    assert_close(loss, loss_ref.detach(), 1e-5, "mpc loss")
    assert abs(float(metrics["AverageLoss"]) - float(loss_ref)) < 1e-4 * max(1.0, float(loss_ref))
    loss.backward()
    for name, t in p.items():
        if name.endswith(om.TRAINABLE_SKIP) or name.endswith(ZERO_GRAD):
            continue
        gref = t.grad if t.grad is not None else torch.zeros_like(t)
        assert rel_err(model.store.grad_view(name), gref) <= 1e-4, name


def test_mpc_trains_under_graph_replay_and_hands_weights_to_the_speech_transformer():
    """A few optimizer steps of the MPC model through the solver (HIP-graph replay included), then
    restore_from_pretrained_model(..., "mpc") (speech_transformer.py:240-245): x_net and encoder of
    the ASR model equal the pre-trained ones, everything else keeps its own initial values."""
    from athena_amd import BaseSolver, MtlTransformerCtc, WarmUpAdam
    from athena_amd.data import model_config
    cfg = dict(MPC_CFG, rate=0.1)
    model = _mpc_model("bf16", cfg=cfg)
    solver = BaseSolver(model, WarmUpAdam({"d_model": 32, "warmup_steps": 10, "k": 0.2}),
                        config={"clip_norm": 1.0, "enable_tf_function": True, "log_interval": 10 ** 9})
    s = model.prepare_samples(_batch())
    losses = [float(solver.train_step(s)[0]) for _ in range(6)]
    assert all(l == l and l > 0 for l in losses) and losses[-1] < losses[0]
    assert abs(float(model.metric.result()) - sum(losses) / len(losses)) < 1e-3 * losses[0]

    asr = MtlTransformerCtc(DataDescriptions(20, 12), model_config(32, 16, 4, 2, 1, 64), precision="bf16", seed=9)
    before = {k: v.clone() for k, v in asr.store.state_dict().items()}
    asr.restore_from_pretrained_model(model, "mpc")
    mine, theirs = asr.store.state_dict(), model.store.state_dict()
    moved = 0
    for k, v in mine.items():
        if k.startswith(("x_net.", "encoder.")):
            assert torch.equal(v, theirs[k]), k
            moved += 1
        else:
            assert torch.equal(v, before[k]), k
    assert moved > 20
    # the restored model runs a step with the new weights in its bf16 copies
    ds = asr.prepare_samples(om.make_samples(2, 50, 12, 21, 4, seed=2))
    loss, _ = asr.get_loss(asr(ds, training=True), ds, training=True)
    assert torch.isfinite(loss)
