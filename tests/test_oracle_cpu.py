"""The CPU oracle against everything that pins it (no GPU needed).

The reference holds no golden vectors for this path and TensorFlow is not installable here
(parity unpinned), so the oracle is tied down from several independent sides instead: exhaustive
alignment enumeration (committed known-answer file), two separately written CTC recursions, torch's
own CPU kernels, and analytic properties."""
import json
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import ctc as oc
from oracle import model as om

HERE = os.path.dirname(os.path.abspath(__file__))


def _kat():
    return json.load(open(os.path.join(HERE, "golden", "ctc_kat.json")))["cases"]


@pytest.mark.parametrize("idx", range(4))
def test_ctc_known_answers(idx):
    cases = _kat()
    if idx >= len(cases):
        pytest.skip("fewer cases in the file")
    c = cases[idx]
    logits = np.array(c["logits"])
    want = np.array(c["loss"])
    tf_loss, tf_grad = oc.ctc_loss_tf_dense(logits, c["labels"], c["label_length"], c["logit_length"])
    tb_loss, tb_grad = oc.ctc_loss_textbook(logits, c["labels"], c["label_length"], c["logit_length"])
    np.testing.assert_allclose(tf_loss, want, rtol=1e-12)
    np.testing.assert_allclose(tb_loss, want, rtol=1e-12)
    np.testing.assert_allclose(tf_grad, tb_grad, atol=1e-12)
    # the enumeration itself, regenerated
    for b in range(logits.shape[0]):
        bf = oc.ctc_brute_force(logits[b, :c["logit_length"][b]], c["labels"][b][:c["label_length"][b]],
                                logits.shape[2] - 1)
        assert abs(bf - want[b]) < 1e-12


def test_ctc_against_torch_and_gradient_properties():
    rng = np.random.default_rng(7)
    B, T, V, U = 4, 23, 9, 6
    logits = rng.normal(size=(B, T, V)) * 1.5
    labels = rng.integers(0, V - 1, size=(B, U))
    llen = np.array([6, 3, 1, 5])
    tlen = np.array([23, 17, 9, 20])
    labels[0, 2] = labels[0, 1]                       # a repeat needs a blank in between
    for b in range(B):
        labels[b, llen[b]:] = 0
    loss, grad = oc.ctc_loss_tf_dense(logits, labels, llen, tlen)
    lt = torch.tensor(logits, requires_grad=True)
    per = F.ctc_loss(F.log_softmax(lt, -1).transpose(0, 1), torch.tensor(labels), torch.tensor(tlen),
                     torch.tensor(llen), blank=V - 1, reduction="none")
    per.sum().backward()
    np.testing.assert_allclose(loss, per.detach().numpy(), rtol=1e-10)
    np.testing.assert_allclose(grad, lt.grad.numpy(), atol=1e-10)
    # softmax - occupancy: rows sum to zero, nothing beyond logit_length
    assert np.abs(grad.sum(-1)).max() < 1e-10
    for b in range(B):
        assert not np.any(grad[b, tlen[b]:])
    # the mean-over-batch wrapper the model uses
    s = {"output": torch.tensor(labels, dtype=torch.int32), "output_length": torch.tensor(llen, dtype=torch.int32)}
    m = om.ctc_loss_mean(torch.tensor(logits), s, torch.tensor(tlen))
    assert abs(float(m) - loss.mean()) < 1e-10
    mean, mgrad = oc.ctc_mean_loss(logits, labels, llen, tlen)
    assert abs(mean - loss.mean()) < 1e-12
    np.testing.assert_allclose(mgrad, grad / B, atol=1e-12)


def test_ctc_infeasible_is_large_and_finite():
    logits = np.zeros((1, 2, 4))
    loss, grad = oc.ctc_loss_tf_dense(logits, [[1, 1, 2]], [3], [2])        # 3 labels cannot fit 2 frames
    assert np.isfinite(loss).all() and loss[0] > 500
    assert np.isfinite(grad).all()


def test_conv_same_padding_matches_explicit_padding():
    g = torch.Generator().manual_seed(0)
    for H, W in [(12, 10), (13, 7), (5, 6), (1, 1)]:
        x = torch.randn(2, H, W, 3, generator=g, dtype=torch.float64)
        w = torch.randn(3, 3, 3, 5, generator=g, dtype=torch.float64)
        got = om.conv2d_same_s2(x, w)
        # TF "same": total = max((out-1)*2+3-in, 0), before = total // 2
        def pads(n):
            o = (n + 1) // 2
            t = max((o - 1) * 2 + 3 - n, 0)
            return t // 2, t - t // 2
        (pt, pb), (pl, pr) = pads(H), pads(W)
        xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
        ref = F.conv2d(xp, w.permute(3, 2, 0, 1), stride=2).permute(0, 2, 3, 1)
        assert got.shape == (2, (H + 1) // 2, (W + 1) // 2, 5)
        assert float((got - ref).abs().max()) < 1e-12


def test_norm_layers_and_activations_match_torch():
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 7, 16, generator=g, dtype=torch.float64)
    ga, be = torch.rand(16, generator=g, dtype=torch.float64) + 0.5, torch.randn(16, generator=g, dtype=torch.float64)
    assert float((om.layer_norm(x, ga, be, 1e-8) - F.layer_norm(x, (16,), ga, be, 1e-8)).abs().max()) < 1e-12
    x4 = torch.randn(3, 5, 4, 16, generator=g, dtype=torch.float64)
    ref = F.batch_norm(x4.permute(0, 3, 1, 2), None, None, ga, be, True, 0.0, 1e-3).permute(0, 2, 3, 1)
    assert float((om.batch_norm_train(x4, ga, be, 1e-3)[0] - ref).abs().max()) < 1e-12
    mm, mv = torch.randn(16, generator=g, dtype=torch.float64), torch.rand(16, generator=g, dtype=torch.float64) + 0.1
    ref = F.batch_norm(x4.permute(0, 3, 1, 2), mm, mv, ga, be, False, 0.0, 1e-3).permute(0, 2, 3, 1)
    assert float((om.batch_norm_infer(x4, ga, be, mm, mv, 1e-3) - ref).abs().max()) < 1e-12
    assert float((om.gelu_tanh(x) - F.gelu(x, approximate="tanh")).abs().max()) < 1e-12
    assert float((om.relu6(x * 5) - F.relu6(x * 5)).abs().max()) == 0.0


def test_positional_encoding_formula():
    pe = om.make_positional_encoding(50, 16)
    pe = pe.reshape(-1, 16)
    for pos in (0, 1, 7, 49):
        for i in (0, 1, 6, 15):
            ang = pos / (10000 ** (2 * (i // 2) / 16))
            want = math.sin(ang) if i % 2 == 0 else math.cos(ang)
            assert abs(float(pe[pos, i]) - want) < 1e-6


def test_label_smoothed_ce_against_torch():
    g = torch.Generator().manual_seed(3)
    B, U, V = 3, 5, 11
    logits = torch.randn(B, U + 1, V, generator=g, dtype=torch.float64)
    labels = torch.randint(1, V - 1, (B, U), generator=g, dtype=torch.int32)
    lens = torch.tensor([5, 2, 4], dtype=torch.int32)
    for b in range(B):
        labels[b, lens[b]:] = 0
    s = {"output": labels, "output_length": lens}
    for ls in (0.0, 0.1):
        got = om.seq2seq_ce_loss(logits, s, V, V - 1, ls)
        tgt = om.insert_eos_in_labels(labels, V - 1, lens).long()
        per = F.cross_entropy(logits.reshape(-1, V), tgt.reshape(-1), reduction="none", label_smoothing=ls)
        per = per.reshape(B, U + 1) * (tgt != 0)
        assert abs(float(got) - float(per.sum(-1).mean())) < 1e-12
    tok = om.seq2seq_ce_loss(logits, s, V, V - 1, 0.0, by_token=True)
    assert abs(float(tok) - float(per_sum_tokens(logits, tgt))) < 1e-12


def per_sum_tokens(logits, tgt):
    V = logits.shape[-1]
    per = F.cross_entropy(logits.reshape(-1, V), tgt.reshape(-1), reduction="none").reshape(tgt.shape) * (tgt != 0)
    return per.sum() / (tgt != 0).sum()


def test_masks_lengths_and_label_helpers():
    assert om.compute_logit_length(torch.tensor([1000, 999, 7, 1])).tolist() == [250, 250, 2, 1]
    lab = torch.tensor([[3, 4, 0], [5, 0, 0]], dtype=torch.int32)
    assert om.insert_sos_in_labels(lab, 9).tolist() == [[9, 3, 4, 0], [9, 5, 0, 0]]
    assert om.insert_eos_in_labels(lab, 9, torch.tensor([2, 1])).tolist() == [[3, 4, 9, 0], [5, 9, 0, 0]]
    y0 = om.insert_sos_in_labels(lab, 9)
    in_mask, out_mask = om.create_multihead_mask(5, torch.tensor([5, 3]), y0, torch.float64)
    assert in_mask.shape == (2, 1, 1, 5) and in_mask[1, 0, 0].tolist() == [0, 0, 0, 1, 1]
    assert out_mask.shape == (2, 1, 4, 4)
    assert out_mask[0, 0, 0].tolist() == [0, 1, 1, 1]          # causal
    assert out_mask[1, 0, 3].tolist() == [0, 0, 1, 1]          # label id 0 is masked as a key


def test_edit_distance_and_greedy_decode():
    assert om.edit_distance([1, 2, 3], [1, 2, 3]) == 0
    assert om.edit_distance([], [1, 2]) == 2
    assert om.edit_distance([1, 2, 3, 4], [1, 3, 4, 5]) == 2
    V = 5
    ids = torch.tensor([[1, 1, 4, 1, 2, 2, 4, 4, 3]])
    logits = F.one_hot(ids, V).double() * 10
    assert om.ctc_greedy_decode(logits, torch.tensor([9])) == [[1, 1, 2, 3]]
    assert om.ctc_greedy_decode(logits, torch.tensor([5])) == [[1, 1, 2]]


def test_optimizer_restatement():
    # lr = k * d^-0.5 * min(step^-0.5, step * warmup^-1.5); the first update (step 0) has lr 0
    assert om.warmup_lr(0, 256, 4000, 1.0) == 0.0
    assert abs(om.warmup_lr(4000, 256, 4000, 1.0) - 256 ** -0.5 * 4000 ** -0.5) < 1e-15
    assert abs(om.warmup_lr(10, 256, 4000, 0.5) - 0.5 * 256 ** -0.5 * 10 * 4000 ** -1.5) < 1e-15
    assert abs(om.warmup_lr(200, 64, 10, 1.0, decay_steps=100, decay_rate=0.5) -
               0.25 * 64 ** -0.5 * 200 ** -0.5) < 1e-15
    g = torch.tensor([3.0, 4.0], dtype=torch.float64)
    assert torch.allclose(om.clip_by_norm(g, 10.0), g)
    assert torch.allclose(om.clip_by_norm(g, 1.0), g / 5)
    # one Adam step against torch.optim.Adam (Keras epsilon placement differs only by eps scaling at step 1)
    p = torch.tensor([1.0, -2.0], dtype=torch.float64)
    m, v = torch.zeros(2, dtype=torch.float64), torch.zeros(2, dtype=torch.float64)
    p2 = p.clone()
    om.adam_update(p2, g, m, v, 0, 0.1)
    # Keras: lr_t = lr * sqrt(1-b2^t)/(1-b1^t); p -= lr_t * m / (sqrt(v) + eps)
    lr_t = 0.1 * math.sqrt(1 - 0.999) / (1 - 0.9)
    want = p - lr_t * (0.1 * g) / (torch.sqrt(0.001 * g * g) + 1e-7)
    assert torch.allclose(p2, want, atol=1e-12)


def test_train_step_decreases_loss_and_is_deterministic():
    cfg = dict(d_model=32, num_filters=8, num_heads=2, num_encoder_layers=1, num_decoder_layers=1, dff=32, rate=0.0,
               label_smoothing_rate=0.1, num_class=8, feat_dim=8, mtl_weight=0.5, return_encoder_output=True)
    runs = []
    for _ in range(2):
        p = om.init_params(cfg, seed=3, dtype=torch.float64)
        for k, v in p.items():
            if not k.endswith(om.TRAINABLE_SKIP):
                v.requires_grad_(True)
        s = om.make_samples(2, 24, 8, 8, 3, seed=4, dtype=torch.float64)
        st = {"iterations": 0, "m": {}, "v": {}}
        losses = [float(om.train_step(p, st, s, cfg, {"clip_norm": 100.0},
                                      dict(d_model=32, warmup_steps=4, k=0.05))[0]) for _ in range(12)]
        runs.append(losses)
    assert runs[0] == runs[1]
    assert runs[0][-1] < runs[0][1]          # (step 0 has lr 0: losses[0] == losses[1])
    assert abs(runs[0][0] - runs[0][1]) < 1e-12


def test_mpc_loss_oracle_against_explicit_loops():
    """oracle.model.mpc_loss (reference athena/loss.py:91-105) against the definition written out:
    sum over features of |target - output| where the input is 0 and the frame is inside the
    utterance, averaged over all batch x frame positions (padded ones included in the count)."""
    from oracle import model as om
    g = torch.Generator().manual_seed(3)
    B, Tp, N = 2, 5, 6
    logits = torch.randn(B, Tp, N, generator=g, dtype=torch.float64)
    target = torch.randn(B, Tp, N, generator=g, dtype=torch.float64)
    masked = target.clone()
    masked[0, 1] = 0.0
    masked[1, 0] = 0.0
    masked[1, 4] = 0.0          # beyond the utterance: not counted
    masked[0, 3, 2] = 0.0       # a single feature that happens to be 0 counts as well
    length = torch.tensor([5, 3])
    total = 0.0
    for b in range(B):
        for t in range(Tp):
            if t >= int(length[b]):
                continue
            for c in range(N):
                if float(masked[b, t, c]) == 0.0:
                    total += abs(float(target[b, t, c]) - float(logits[b, t, c]))
    got = om.mpc_loss(logits, {"output": target.reshape(B, Tp * 2, 3), "input": masked.reshape(B, Tp * 2, 3, 1)}, length)
    assert abs(float(got) - total / (B * Tp)) < 1e-12


# ---------------------------------------------------------------- DeepSpeech oracle (oracle/deep_speech.py)
def test_oracle_gru_and_lstm_layers_match_torch_nn():
    """The oracle's Keras-order GRU (reset_after) and LSTM against torch.nn.GRU / torch.nn.LSTM, which
    implement the same cell equations with the gates in another order (GRU: r, z, n; LSTM: i, f, g, o)
    and two bias vectors."""
    from oracle import deep_speech as ods
    g = torch.Generator().manual_seed(5)
    B, T, K, H = 3, 7, 6, 5
    x = torch.randn(B, T, K, generator=g, dtype=torch.float64)
    # GRU
    W = torch.randn(K, 3 * H, generator=g, dtype=torch.float64) * 0.5
    U = torch.randn(H, 3 * H, generator=g, dtype=torch.float64) * 0.5
    b = torch.randn(2, 3 * H, generator=g, dtype=torch.float64) * 0.3
    ref = torch.nn.GRU(K, H, batch_first=True).double()
    perm = torch.cat([torch.arange(H, 2 * H), torch.arange(0, H), torch.arange(2 * H, 3 * H)])   # keras z r h -> r z n
    with torch.no_grad():
        ref.weight_ih_l0.copy_(W.t()[perm])
        ref.weight_hh_l0.copy_(U.t()[perm])
        ref.bias_ih_l0.copy_(b[0][perm])
        ref.bias_hh_l0.copy_(b[1][perm])
    want, _ = ref(x)
    got = ods.gru_layer(x, W, U, b)
    assert float((got - want).abs().max()) < 1e-12
    # LSTM (same gate order; torch adds its two biases)
    W = torch.randn(K, 4 * H, generator=g, dtype=torch.float64) * 0.5
    U = torch.randn(H, 4 * H, generator=g, dtype=torch.float64) * 0.5
    b = torch.randn(4 * H, generator=g, dtype=torch.float64) * 0.3
    ref = torch.nn.LSTM(K, H, batch_first=True).double()
    with torch.no_grad():
        ref.weight_ih_l0.copy_(W.t())
        ref.weight_hh_l0.copy_(U.t())
        ref.bias_ih_l0.copy_(b)
        ref.bias_hh_l0.zero_()
    want, _ = ref(x)
    got = ods.lstm_layer(x, W, U, b)
    assert float((got - want).abs().max()) < 1e-12


def test_oracle_conv2d_same_matches_explicit_sum():
    """"same" padding with even and odd sizes and both stride pairs of the model, against the defining
    sum written out with loops."""
    from oracle import deep_speech as ods
    g = torch.Generator().manual_seed(6)
    for (H, W, kh, kw, sh, sw) in ((9, 6, 5, 3, 2, 2), (8, 5, 4, 3, 2, 1), (7, 7, 41, 11, 2, 2)):
        x = torch.randn(2, H, W, 2, generator=g, dtype=torch.float64)
        w = torch.randn(kh, kw, 2, 3, generator=g, dtype=torch.float64)
        got = ods.conv2d_same(x, w, (sh, sw))
        Ho, Wo = -(-H // sh), -(-W // sw)
        assert tuple(got.shape) == (2, Ho, Wo, 3)
        ph, _ = ods.same_pad(H, kh, sh)
        pw, _ = ods.same_pad(W, kw, sw)
        want = torch.zeros(2, Ho, Wo, 3, dtype=torch.float64)
        for ho in range(Ho):
            for wo in range(Wo):
                for a in range(kh):
                    for c in range(kw):
                        h, v = sh * ho - ph + a, sw * wo - pw + c
                        if 0 <= h < H and 0 <= v < W:
                            want[:, ho, wo] += x[:, h, v] @ w[a, c]
        assert float((got - want).abs().max()) < 1e-10


def test_oracle_deep_speech_forward_shapes_and_loss():
    from oracle import deep_speech as ods
    from oracle import model as om
    cfg = {"conv_filters": 8, "rnn_hidden_size": 16, "num_rnn_layers": 2, "rnn_type": "gru"}
    p = ods.init_params(cfg, 10, 12, seed=2)
    s = om.make_samples(2, 37, 10, 12, 3, seed=3, full_length=False, dtype=torch.float64)
    loss, logits, grads = ods.loss_and_grads(p, s, cfg)
    assert tuple(logits.shape) == (2, 10, 12) and torch.isfinite(loss) and float(loss) > 0
    assert set(grads) == {k for k in p if not k.endswith(("moving_mean", "moving_variance"))}
    assert all(torch.isfinite(v).all() for v in grads.values())
