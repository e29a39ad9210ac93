"""Oracle (test infrastructure): the DeepSpeech CTC model on torch-CPU.

A functional restatement of the reference's athena/models/deep_speech.py:45-96 in float64 (or
float32); torch autograd supplies the backward pass.  Parameters live in a flat ``dict`` under the
names Keras gives the layers in creation order (conv2d, batch_normalization, conv2d_1,
batch_normalization_1, rnn, batch_normalization_2, rnn_1, ..., dense, dense_1), in Keras shapes:
Conv2D kernels HWIO, GRUCell kernel [in, 3H] / recurrent_kernel [H, 3H] / bias [2, 3H], LSTMCell
kernel [in, 4H] / recurrent_kernel [H, 4H] / bias [4H], Dense kernels [in, out].

TensorFlow/Keras semantics (third-party, not installed here) are restated from their published
behaviour: Conv2D "same" padding puts total//2 before and the rest after, with total =
max((ceil(n/s)-1)*s + k - n, 0); GRUCell (TF 2 default reset_after=True, gate order z, r, h):
  z = sigmoid(x Wz + bz + h Uz + cz), r = sigmoid(x Wr + br + h Ur + cr),
  hh = tanh(x Wh + bh + r * (h Uh + ch)), h' = z * h + (1 - z) * hh          (b = bias[0], c = bias[1]);
LSTMCell (gate order i, f, c, o): c' = f * c + i * tanh(.), h' = o * tanh(c'); the RNN wrapper starts
from zero states, sees no mask (the model passes none) and returns every step's h.
PARITY UNPINNED against TensorFlow itself; tests/test_oracle_cpu.py ties the two cells to torch.nn.GRU
and torch.nn.LSTM (independent implementations of the same equations) with the gates reordered.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from .model import batch_norm_infer, batch_norm_train, ctc_loss_mean, relu6


def same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def conv2d_same(x_nhwc, w_hwio, strides):
    """Keras Conv2D(padding="same", use_bias=False), channels last (deep_speech.py:49-56,59-66)."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    x = x_nhwc.permute(0, 3, 1, 2)
    pt, pb = same_pad(x.shape[2], kh, strides[0])
    pl, pr = same_pad(x.shape[3], kw, strides[1])
    x = F.pad(x, (pl, pr, pt, pb))
    y = F.conv2d(x, w_hwio.permute(3, 2, 0, 1), stride=strides)
    return y.permute(0, 2, 3, 1)


def gru_layer(x, kernel, recurrent, bias):
    """tf.keras.layers.RNN([GRUCell(H)], return_sequences=True) on x [B, T, in]."""
    B, T, _ = x.shape
    H = recurrent.shape[0]
    xw = x @ kernel + bias[0]
    h = torch.zeros((B, H), dtype=x.dtype)
    out = []
    for t in range(T):
        hu = h @ recurrent + bias[1]
        z = torch.sigmoid(xw[:, t, :H] + hu[:, :H])
        r = torch.sigmoid(xw[:, t, H:2 * H] + hu[:, H:2 * H])
        hh = torch.tanh(xw[:, t, 2 * H:] + r * hu[:, 2 * H:])
        h = z * h + (1 - z) * hh
        out.append(h)
    return torch.stack(out, dim=1)


def lstm_layer(x, kernel, recurrent, bias):
    """tf.keras.layers.RNN([LSTMCell(H)], return_sequences=True) on x [B, T, in]."""
    B, T, _ = x.shape
    H = recurrent.shape[0]
    xw = x @ kernel + bias
    h = torch.zeros((B, H), dtype=x.dtype)
    c = torch.zeros((B, H), dtype=x.dtype)
    out = []
    for t in range(T):
        a = xw[:, t] + h @ recurrent
        i, f = torch.sigmoid(a[:, :H]), torch.sigmoid(a[:, H:2 * H])
        g, o = torch.tanh(a[:, 2 * H:3 * H]), torch.sigmoid(a[:, 3 * H:])
        c = f * c + i * g
        h = o * torch.tanh(c)
        out.append(h)
    return torch.stack(out, dim=1)


def rnn_name(i):
    return "rnn" if i == 0 else "rnn_%d" % i


def bn_name(i):
    return "batch_normalization" if i == 0 else "batch_normalization_%d" % i


def _bn(p, name, x, training, stats_out):
    if training:
        shape = x.shape
        y, mean, var = batch_norm_train(x.reshape(1, 1, -1, shape[-1]), p[name + ".gamma"], p[name + ".beta"])
        if stats_out is not None:
            stats_out[name] = (mean.detach(), var.detach(), x.numel() // shape[-1])
        return y.reshape(shape)
    return batch_norm_infer(x, p[name + ".gamma"], p[name + ".beta"], p[name + ".moving_mean"],
                            p[name + ".moving_variance"])


def forward(p, samples, cfg, training=True, bn_stats_out=None):
    """DeepSpeechModel.call (deep_speech.py:86-88): logits [B, T', num_classes]."""
    x = samples["input"]
    x = relu6(_bn(p, bn_name(0), conv2d_same(x, p["conv2d.kernel"], (2, 2)), training, bn_stats_out))
    x = relu6(_bn(p, bn_name(1), conv2d_same(x, p["conv2d_1.kernel"], (2, 1)), training, bn_stats_out))
    B, T2, F2, C = x.shape
    x = x.reshape(B, T2, F2 * C)
    layer = lstm_layer if "lstm" in cfg["rnn_type"] else gru_layer
    for i in range(cfg["num_rnn_layers"]):
        n = rnn_name(i)
        x = layer(x, p[n + ".kernel"], p[n + ".recurrent_kernel"], p[n + ".bias"])
        x = _bn(p, bn_name(i + 2), x, training, bn_stats_out)
    x = relu6(x @ p["dense.kernel"] + p["dense.bias"])
    return x @ p["dense_1.kernel"] + p["dense_1.bias"]


def compute_logit_length(input_length):
    """ceil(ceil(len / 2) / 2) (deep_speech.py:90-96)."""
    n = input_length.to(torch.float64)
    return torch.ceil(torch.ceil(n / 2) / 2).to(torch.int32)


def get_loss(logits, samples):
    """CTCLoss(blank_index=-1): mean over the batch (deep_speech.py:40, athena/loss.py:35-46)."""
    return ctc_loss_mean(logits, samples, compute_logit_length(samples["input_length"]))


def init_params(cfg, feat_dim, num_classes, seed=1, dtype=torch.float64, in_channels=1):
    """Random parameters in Keras shapes (scaled normal; the product's own initial values are what
    the parity tests copy over, this is for the oracle's own tests)."""
    g = torch.Generator().manual_seed(seed)
    C, H = cfg["conv_filters"], cfg["rnn_hidden_size"]
    G = 4 if "lstm" in cfg["rnn_type"] else 3

    def rnd(*shape, fan=None):
        fan = fan or shape[0]
        return torch.randn(shape, generator=g, dtype=torch.float64) / math.sqrt(fan)

    p = {"conv2d.kernel": rnd(41, 11, in_channels, C, fan=41 * 11 * in_channels),
         "conv2d_1.kernel": rnd(21, 11, C, C, fan=21 * 11 * C)}
    widths = [C] * 2 + [H] * cfg["num_rnn_layers"]
    for i, w in enumerate(widths):
        p[bn_name(i) + ".gamma"] = 1.0 + 0.1 * torch.randn(w, generator=g, dtype=torch.float64)
        p[bn_name(i) + ".beta"] = 0.1 * torch.randn(w, generator=g, dtype=torch.float64)
        p[bn_name(i) + ".moving_mean"] = torch.zeros(w, dtype=torch.float64)
        p[bn_name(i) + ".moving_variance"] = torch.ones(w, dtype=torch.float64)
    width = -(-feat_dim // 2) * C
    for i in range(cfg["num_rnn_layers"]):
        n = rnn_name(i)
        p[n + ".kernel"] = rnd(width, G * H)
        p[n + ".recurrent_kernel"] = rnd(H, G * H)
        p[n + ".bias"] = 0.1 * torch.randn((G * H,) if G == 4 else (2, G * H), generator=g, dtype=torch.float64)
        width = H
    p["dense.kernel"] = rnd(width, H)
    p["dense.bias"] = 0.1 * torch.randn(H, generator=g, dtype=torch.float64)
    p["dense_1.kernel"] = rnd(H, num_classes)
    p["dense_1.bias"] = torch.zeros(num_classes, dtype=torch.float64)
    return {k: v.to(dtype) for k, v in p.items()}


def loss_and_grads(p, samples, cfg):
    """(loss, logits, {name: gradient}) of one training-mode forward pass."""
    leaves = {k: v.clone().requires_grad_(not k.endswith(("moving_mean", "moving_variance"))) for k, v in p.items()}
    logits = forward(leaves, samples, cfg, training=True)
    loss = get_loss(logits, samples)
    names = [k for k, v in leaves.items() if v.requires_grad]
    grads = torch.autograd.grad(loss, [leaves[k] for k in names])
    return loss.detach(), logits.detach(), dict(zip(names, grads))


__all__ = ["forward", "get_loss", "loss_and_grads", "init_params", "gru_layer", "lstm_layer", "conv2d_same",
           "compute_logit_length", "np"]
