"""Oracle (test infrastructure): the MtlTransformerCtc training step on torch-CPU.

A functional, op-for-op restatement of the reference's hot path in float64 or
float32 on the CPU; torch autograd supplies the backward pass, so gradients
are those of exactly this forward.  Parameters live in a flat ``dict`` whose
names and shapes mirror the Keras variables of the reference (Dense kernels
``[in, out]``, Conv2D kernels HWIO, Embedding ``[V, d]``).

TensorFlow/Keras op semantics (third-party, not installed here) are restated
from their published behaviour: Conv2D "same" padding with stride 2 pads
``total//2`` before and the rest after; BatchNormalization in training mode
normalises with the batch mean and biased variance (epsilon 1e-3) and updates
the moving variance with the unbiased estimate (momentum 0.99);
LayerNormalization uses the biased variance with the configured epsilon.
PARITY UNPINNED against TensorFlow itself (see package docstring).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# layers
# ----------------------------------------------------------------------------
def make_positional_encoding(position, d_model):
    """Sinusoid table [position, d_model], float32 (reference athena/layers/functional.py:25-38)."""
    pos = np.arange(position)[:, np.newaxis]
    i = np.arange(d_model)[np.newaxis, :]
    angle_rates = 1 / np.power(10000, (2 * (i // 2)) / np.float32(d_model))
    angle_rads = pos * angle_rates
    angle_rads[:, 0::2] = np.sin(angle_rads[:, 0::2])
    angle_rads[:, 1::2] = np.cos(angle_rads[:, 1::2])
    return torch.from_numpy(angle_rads.astype(np.float32))


def gelu_tanh(x):
    """tanh-form GELU (reference athena/layers/functional.py:101-112)."""
    return x * 0.5 * (1.0 + torch.tanh(math.sqrt(2 / math.pi) * (x + 0.044715 * x ** 3)))


def relu6(x):
    return torch.clamp(x, 0.0, 6.0)


def _same_pad(n, k=3, s=2):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def conv2d_same_s2(x_nhwc, w_hwio):
    """Keras Conv2D(kernel 3x3, strides 2, padding "same", no bias), NHWC in/out
    (reference athena/models/speech_transformer.py:77-84,87-94)."""
    x = x_nhwc.permute(0, 3, 1, 2)
    pt, pb = _same_pad(x.shape[2])
    pl, pr = _same_pad(x.shape[3])
    x = F.pad(x, (pl, pr, pt, pb))
    y = F.conv2d(x, w_hwio.permute(3, 2, 0, 1), stride=2)
    return y.permute(0, 2, 3, 1)


def batch_norm_train(x_nhwc, gamma, beta, eps=1e-3):
    """Keras BatchNormalization() forward in training mode; returns y and the
    batch statistics (mean, biased variance) (speech_transformer.py:85,95)."""
    mean = x_nhwc.mean(dim=(0, 1, 2))
    var = x_nhwc.var(dim=(0, 1, 2), unbiased=False)
    y = (x_nhwc - mean) / torch.sqrt(var + eps) * gamma + beta
    return y, mean, var


def batch_norm_infer(x_nhwc, gamma, beta, moving_mean, moving_var, eps=1e-3):
    return (x_nhwc - moving_mean) / torch.sqrt(moving_var + eps) * gamma + beta


def layer_norm(x, gamma, beta, eps=1e-8):
    """Keras LayerNormalization(epsilon=1e-8) (reference athena/layers/transformer.py:286-288)."""
    mean = x.mean(dim=-1, keepdim=True)
    var = x.var(dim=-1, unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * gamma + beta


def dense(x, p, name):
    return x @ p[name + ".kernel"] + p[name + ".bias"]


def multi_head_attention(p, name, v_in, k_in, q_in, mask, num_heads, unidirectional=False, look_ahead=0):
    """MultiHeadAttention.call(v, k, q, mask) (reference athena/layers/attention.py:147-171)
    with ScaledDotProductAttention (attention.py:47-70).  ``mask`` is float, 1.0 = masked,
    broadcastable to [B, h, Tq, Tk].  ``unidirectional`` adds (1 - band_part(ones, -1, look_ahead))
    * -1e9 first (attention.py:55-58)."""
    B = q_in.shape[0]
    d = p[name + ".wq.kernel"].shape[1]
    depth = d // num_heads

    def split(x):
        return x.reshape(B, -1, num_heads, depth).permute(0, 2, 1, 3)

    q = split(dense(q_in, p, name + ".wq"))
    k = split(dense(k_in, p, name + ".wk"))
    v = split(dense(v_in, p, name + ".wv"))
    logits = q @ k.transpose(-1, -2) / math.sqrt(depth)
    if unidirectional:
        band = torch.tril(torch.ones(logits.shape[-2:], dtype=logits.dtype), diagonal=look_ahead)
        logits = logits + (1 - band) * -1e9
    if mask is not None:
        logits = logits + mask * -1e9
    w = torch.softmax(logits, dim=-1)
    out = (w @ v).permute(0, 2, 1, 3).reshape(B, -1, d)
    return dense(out, p, name + ".dense"), w


# ACTIVATIONS of athena/layers/commons.py:494-506 (without glu, which halves its input's width)
FFN_ACTIVATIONS = {
    "gelu": lambda x: gelu_tanh(x), "relu": F.relu, "relu6": lambda x: relu6(x), "elu": F.elu, "selu": F.selu,
    "leaky_relu": lambda x: F.leaky_relu(x, 0.2), "sigmoid": torch.sigmoid, "softplus": F.softplus,
    "softsign": F.softsign, "tanh": torch.tanh,
}


def ffn(p, name, x, rate, training, activation="gelu"):
    """Dense(dff, activation) -> Dropout -> Dense(d) -> Dropout (transformer.py:262-282)."""
    h = FFN_ACTIVATIONS[activation](dense(x, p, name + ".0"))
    h = F.dropout(h, rate, training)
    h = dense(h, p, name + ".1")
    return F.dropout(h, rate, training)


def encoder_layer(p, name, src, src_mask, cfg, training):
    """TransformerEncoderLayer.call (reference athena/layers/transformer.py:296-310)."""
    a, _ = multi_head_attention(p, name + ".self_attn", src, src, src, src_mask, cfg["num_heads"],
                                cfg.get("unidirectional", False), cfg.get("look_ahead", 0))
    out = layer_norm(src + F.dropout(a, cfg["rate"], training),
                     p[name + ".norm1.gamma"], p[name + ".norm1.beta"])
    out = layer_norm(out + ffn(p, name + ".ffn", out, cfg["rate"], training, cfg.get("ffn_activation", "gelu")),
                     p[name + ".norm2.gamma"], p[name + ".norm2.beta"])
    return out


def decoder_layer(p, name, tgt, memory, tgt_mask, memory_mask, cfg, training):
    """TransformerDecoderLayer.call (reference athena/layers/transformer.py:374-388)."""
    a, _ = multi_head_attention(p, name + ".attn1", tgt, tgt, tgt, tgt_mask, cfg["num_heads"])
    out = layer_norm(tgt + F.dropout(a, cfg["rate"], training),
                     p[name + ".norm1.gamma"], p[name + ".norm1.beta"])
    a2, w2 = multi_head_attention(p, name + ".attn2", memory, memory, out, memory_mask, cfg["num_heads"])
    out = layer_norm(out + F.dropout(a2, cfg["rate"], training),
                     p[name + ".norm2.gamma"], p[name + ".norm2.beta"])
    out = layer_norm(out + ffn(p, name + ".ffn", out, cfg["rate"], training, cfg.get("ffn_activation", "gelu")),
                     p[name + ".norm3.gamma"], p[name + ".norm3.beta"])
    return out, w2


# ----------------------------------------------------------------------------
# masks / labels (reference athena/utils/misc.py)
# ----------------------------------------------------------------------------
def compute_logit_length(input_length):
    """ceil(ceil(len/2)/2) (reference athena/models/speech_transformer.py:156-162)."""
    x = input_length.to(torch.float32)
    return torch.ceil(torch.ceil(x / 2) / 2).to(torch.int32)


def insert_sos_in_labels(labels, sos):
    """misc.py:32-34"""
    return torch.cat([torch.full_like(labels[:, :1], sos), labels], dim=-1)


def insert_eos_in_labels(labels, eos, lengths):
    """misc.py:50-57"""
    out = torch.cat([labels, torch.zeros_like(labels[:, :1])], dim=-1)
    out = out + F.one_hot(lengths.long(), out.shape[1]).to(out.dtype) * eos
    return out


def create_multihead_mask(x_len_total, x_length, y0, dtype):
    """misc.py:68-90 (reverse=False): float masks, 1.0 = masked."""
    ar = torch.arange(x_len_total)
    x_mask = 1.0 - (ar[None, :] < x_length[:, None]).to(dtype)
    x_mask = x_mask[:, None, None, :]
    y_mask = (y0 == 0).to(dtype)[:, None, None, :]
    L = y0.shape[1]
    look_ahead = 1 - torch.tril(torch.ones(L, L, dtype=dtype))
    y_mask = torch.maximum(y_mask, look_ahead)
    return x_mask, y_mask


# ----------------------------------------------------------------------------
# model
# ----------------------------------------------------------------------------
def x_net(p, x0, cfg, training, bn_stats_out=None):
    """Conv2D-BN-relu6 x2, reshape, Dense+relu6, +PE, dropout (speech_transformer.py:72-105)."""
    h = conv2d_same_s2(x0, p["x_net.conv1.kernel"])
    if training:
        h, m1, v1 = batch_norm_train(h, p["x_net.bn1.gamma"], p["x_net.bn1.beta"])
    else:
        h = batch_norm_infer(h, p["x_net.bn1.gamma"], p["x_net.bn1.beta"],
                             p["x_net.bn1.moving_mean"], p["x_net.bn1.moving_variance"])
    h = relu6(h)
    n1 = h.shape[0] * h.shape[1] * h.shape[2]
    h = conv2d_same_s2(h, p["x_net.conv2.kernel"])
    if training:
        h, m2, v2 = batch_norm_train(h, p["x_net.bn2.gamma"], p["x_net.bn2.beta"])
    else:
        h = batch_norm_infer(h, p["x_net.bn2.gamma"], p["x_net.bn2.beta"],
                             p["x_net.bn2.moving_mean"], p["x_net.bn2.moving_variance"])
    h = relu6(h)
    n2 = h.shape[0] * h.shape[1] * h.shape[2]
    if training and bn_stats_out is not None:
        bn_stats_out.update({"bn1": (m1.detach(), v1.detach(), n1), "bn2": (m2.detach(), v2.detach(), n2)})
    B, T2, F2, C = h.shape
    h = h.reshape(B, T2, F2 * C)
    h = relu6(dense(h, p, "x_net.dense"))
    pe = make_positional_encoding(800, cfg["d_model"]).to(h.dtype)
    h = h + pe[None, :T2, :]
    return F.dropout(h, cfg["rate"], training)


def y_net(p, y0, cfg, training):
    """Embedding, *sqrt(d) + PE, dropout (speech_transformer.py:108-113, commons.py:35-41)."""
    e = p["y_net.embedding"][y0.long()]
    d = cfg["d_model"]
    e = e * torch.sqrt(torch.tensor(float(d), dtype=torch.float32)).to(e.dtype)
    pe = make_positional_encoding(800, d).to(e.dtype)
    e = e + pe[None, :y0.shape[1], :]
    return F.dropout(e, cfg["rate"], training)


def forward(p, samples, cfg, training=True, bn_stats_out=None):
    """MtlTransformerCtc.call (reference athena/models/mtl_seq2seq.py:80-84) over
    SpeechTransformer.call (speech_transformer.py:135-154).
    Returns (attention_logits [B,U+1,V], ctc_logits [B,T',V], encoder_output)."""
    V = cfg["num_class"]
    sos = V - 1
    x0 = samples["input"]
    y0 = insert_sos_in_labels(samples["output"], sos)
    x = x_net(p, x0, cfg, training, bn_stats_out)
    y = y_net(p, y0, cfg, training)
    input_length = compute_logit_length(samples["input_length"])
    input_mask, output_mask = create_multihead_mask(x.shape[1], input_length, y0, x.dtype)
    mem = x
    for i in range(cfg["num_encoder_layers"]):
        mem = encoder_layer(p, "encoder.%d" % i, mem, input_mask, cfg, training)
    out = y
    for i in range(cfg["num_decoder_layers"]):
        out, _ = decoder_layer(p, "decoder.%d" % i, out, mem, output_mask, input_mask, cfg, training)
    att_logits = dense(out, p, "final_layer")
    ctc_logits = dense(mem, p, "ctc_decoder")
    return att_logits, ctc_logits, mem


def mpc_forward(p, samples, cfg, training=True, bn_stats_out=None):
    """MaskedPredictCoding.call (reference athena/models/masked_pc.py:122-136): x_net, the encoder
    stack without any mask, Dense(4 * feature size)."""
    x = x_net(p, samples["input"], cfg, training, bn_stats_out)
    for i in range(cfg["num_encoder_layers"]):
        x = encoder_layer(p, "encoder.%d" % i, x, None, cfg, training)
    return dense(x, p, "final_layer")


def mpc_loss(logits, samples, logit_length):
    """MPCLoss.__call__ (reference athena/loss.py:91-105)."""
    shape = logits.shape
    target = samples["output"].reshape(shape).to(logits.dtype)
    loss = target - logits
    loss = loss * (samples["input"].reshape(shape) == 0).to(loss.dtype)
    seq_mask = (torch.arange(shape[1])[None, :] < logit_length[:, None]).to(loss.dtype)
    loss = loss * seq_mask[:, :, None]
    return loss.abs().sum(-1).mean()


# ----------------------------------------------------------------------------
# losses (reference athena/loss.py:24-80)
# ----------------------------------------------------------------------------
class _TfDenseCtc(torch.autograd.Function):
    """Per-sample loss of oracle.ctc.ctc_loss_tf_dense (TensorFlow's dense branch with its finite
    log(1e-307) "log zero") with that function's own gradient."""

    @staticmethod
    def forward(ctx, logits, labels, label_length, logit_length):
        from . import ctc as octc
        loss, grad = octc.ctc_loss_tf_dense(logits.detach().double().numpy(), labels.numpy(), label_length.numpy(),
                                            logit_length.numpy(), blank_index=-1)
        ctx.grad = torch.from_numpy(grad).to(logits.dtype)
        return torch.from_numpy(loss).to(logits.dtype)

    @staticmethod
    def backward(ctx, g):
        return ctx.grad * g[:, None, None], None, None, None


def ctc_loss_mean(ctc_logits, samples, logit_length):
    """CTCLoss.__call__ (loss.py:35-46): blank = V-1, mean over the batch.  torch's CPU
    ctc_loss is used as the differentiable implementation; tests tie it to
    oracle.ctc (TF-dense restatement, textbook recursion, brute force).  Samples whose labels do
    not fit their frames, where torch returns inf and TensorFlow's dense branch a finite value of
    about 7e2, take the TF-dense restatement instead."""
    V = ctc_logits.shape[-1]
    lp = F.log_softmax(ctc_logits, dim=-1).transpose(0, 1)
    per = F.ctc_loss(lp, samples["output"].long(), logit_length.long(), samples["output_length"].long(),
                     blank=V - 1, reduction="none", zero_infinity=True)
    probe = F.ctc_loss(lp.detach(), samples["output"].long(), logit_length.long(), samples["output_length"].long(),
                       blank=V - 1, reduction="none", zero_infinity=False)
    bad = torch.isinf(probe) | (logit_length <= 0)
    if bool(bad.any()):
        idx = torch.nonzero(bad).flatten()
        tf = _TfDenseCtc.apply(ctc_logits[idx], samples["output"][idx], samples["output_length"][idx].long(),
                               logit_length[idx].long())
        per = per.index_put((idx,), tf)
    return per.mean()


def seq2seq_ce_loss(logits, samples, num_classes, eos, label_smoothing=0.0,
                    by_token=False, by_sequence=True):
    """Seq2SeqSparseCategoricalCrossentropy.__call__ (loss.py:63-80)."""
    labels = insert_eos_in_labels(samples["output"], eos, samples["output_length"])
    mask = (labels != 0).to(logits.dtype)
    y = F.one_hot(labels.long(), num_classes).to(logits.dtype)
    if label_smoothing != 0.0:
        y = (1.0 - label_smoothing) * y + label_smoothing / num_classes      # misc.py:209-217
    seq_len = y.shape[1]
    logits = logits[:, :seq_len, :]
    loss = -(y * F.log_softmax(logits, dim=-1)).sum(-1)
    loss = loss * mask
    if by_token:
        return loss.sum() / mask.sum()
    if by_sequence:
        loss = loss.sum(-1)
    return loss.mean()


def get_loss(att_logits, ctc_logits, samples, cfg):
    """MtlTransformerCtc.get_loss (mtl_seq2seq.py:86-97) without the metrics."""
    V = cfg["num_class"]
    logit_length = compute_logit_length(samples["input_length"])
    ctc = ctc_loss_mean(ctc_logits, samples, logit_length)
    ce = seq2seq_ce_loss(att_logits, samples, V, V - 1, cfg.get("label_smoothing_rate", 0.0))
    w = cfg["mtl_weight"]
    return w * ce + (1.0 - w) * ctc, ce, ctc


# ----------------------------------------------------------------------------
# metrics (reference athena/metrics.py:56-132, athena/utils/misc.py:98-126)
# ----------------------------------------------------------------------------
def edit_distance(a, b):
    """Levenshtein distance between two int sequences (tf.edit_distance, normalize=False)."""
    la, lb = len(a), len(b)
    prev = list(range(lb + 1))
    for i in range(1, la + 1):
        cur = [i] + [0] * lb
        for j in range(1, lb + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (a[i - 1] != b[j - 1]))
        prev = cur
    return prev[lb]


def ctc_greedy_decode(ctc_logits, logit_length):
    """tf.nn.ctc_greedy_decoder (merge_repeated=True, blank = V-1) (metrics.py:127-130)."""
    V = ctc_logits.shape[-1]
    best = ctc_logits.argmax(-1)
    out = []
    for b in range(best.shape[0]):
        seq, prev = [], None
        for t in range(int(logit_length[b])):
            c = int(best[b, t])
            if c != prev and c != V - 1:
                seq.append(c)
            prev = c
        out.append(seq)
    return out


def ctc_accuracy_counts(ctc_logits, samples, logit_length):
    """CTCAccuracy.__call__ (metrics.py:118-132): (num_errs, num_labels) or None when skipped."""
    if ctc_logits.shape[1] <= samples["output"].shape[1] + 1:
        return None
    hyp = ctc_greedy_decode(ctc_logits, logit_length)
    errs, total = 0, 0
    for b, h in enumerate(hyp):
        ref = [int(x) for x in samples["output"][b] if int(x) != 0]
        errs += edit_distance(h, ref)
        total += len(ref)
    return float(errs), float(total)


def seq2seq_accuracy_counts(att_logits, samples, eos):
    """Seq2SeqSparseCategoricalAccuracy.__call__ (metrics.py:101-106) + validate_seqs."""
    pred = att_logits.argmax(-1)
    errs, total = 0, 0
    for b in range(pred.shape[0]):
        seq = []
        for c in pred[b].tolist():
            if c == eos:
                break
            seq.append(c)
        seq = [c for c in seq if c != 0]
        ref = [int(x) for x in samples["output"][b] if int(x) != 0]
        errs += edit_distance(seq, ref)
        total += len(ref)
    return float(errs), float(total)


# ----------------------------------------------------------------------------
# optimizer (reference athena/solver.py:78-99, athena/utils/learning_rate.py:24-86)
# ----------------------------------------------------------------------------
def warmup_lr(step, d_model=512, warmup_steps=8000, k=0.5, decay_steps=100000, decay_rate=1.0):
    """WarmUpLearningSchedule.__call__ (learning_rate.py:52-58); step is optimizer.iterations."""
    step = float(step)
    arg1 = float("inf") if step == 0 else step ** -0.5
    arg2 = step * (warmup_steps ** -1.5)
    kk = k * decay_rate ** (step // decay_steps)
    return kk * d_model ** -0.5 * min(arg1, arg2)


def clip_by_norm(g, clip_norm):
    """tf.clip_by_norm per tensor (solver.py:78-87): g * c / max(||g||, c)."""
    if clip_norm <= 0:
        return g
    n = torch.linalg.vector_norm(g)
    return g * clip_norm / torch.maximum(n, torch.tensor(clip_norm, dtype=g.dtype))


def adam_update(param, g, m, v, iterations, lr, beta1=0.9, beta2=0.999, eps=1e-7):
    """Keras Adam dense update with t = iterations + 1 (learning_rate.py:61-86)."""
    t = iterations + 1
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    lr_t = lr * math.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    param.sub_(lr_t * m / (torch.sqrt(v) + eps))


TRAINABLE_SKIP = ("moving_mean", "moving_variance")


def train_step(p, opt_state, samples, cfg, solver_cfg, opt_cfg, training=True):
    """BaseSolver.train_step (solver.py:89-99).  ``p`` maps names to leaf tensors
    (requires_grad for trainables); ``opt_state`` = {"iterations": int, "m": {}, "v": {}}.
    Returns (loss, ce, ctc, grads-before-clip dict)."""
    for t in p.values():
        if t.grad is not None:
            t.grad = None
    stats = {}
    att, ctc_logits, _ = forward(p, samples, cfg, training=training, bn_stats_out=stats)
    loss, ce, ctc = get_loss(att, ctc_logits, samples, cfg)
    loss.backward()
    grads = {}
    lr = warmup_lr(opt_state["iterations"], **opt_cfg)
    with torch.no_grad():
        for name, t in p.items():
            if name.endswith(TRAINABLE_SKIP) or t.grad is None:
                continue
            grads[name] = t.grad.detach().clone()
            g = clip_by_norm(t.grad, solver_cfg.get("clip_norm", 100.0))
            if name not in opt_state["m"]:
                opt_state["m"][name] = torch.zeros_like(t)
                opt_state["v"][name] = torch.zeros_like(t)
            adam_update(t, g, opt_state["m"][name], opt_state["v"][name], opt_state["iterations"], lr)
        # moving statistics (Keras BN momentum 0.99, unbiased variance for the moving average)
        for bn in ("bn1", "bn2"):
            if bn in stats:
                mean, var, n = stats[bn]
                mm, mv = p["x_net.%s.moving_mean" % bn], p["x_net.%s.moving_variance" % bn]
                mm.mul_(0.99).add_(mean, alpha=0.01)
                mv.mul_(0.99).add_(var * (n / max(n - 1, 1)), alpha=0.01)
    opt_state["iterations"] += 1
    return loss.detach(), ce.detach(), ctc.detach(), grads


# ----------------------------------------------------------------------------
# parameter initialisation (Keras defaults restated; used for synthetic weights)
# ----------------------------------------------------------------------------
def init_params(cfg, seed=1, dtype=torch.float32):
    """Random-init parameters with the reference's initialisers: glorot-uniform for Dense/Conv
    without an explicit initialiser, truncated-normal(0.02) for attention/FFN kernels
    (attention.py:117, transformer.py:267), uniform(-0.05, 0.05) Embedding, zeros bias,
    BN/LN gamma 1 beta 0."""
    g = torch.Generator().manual_seed(seed)
    d, C, V, dff = cfg["d_model"], cfg["num_filters"], cfg["num_class"], cfg["dff"]
    Fdim = cfg["feat_dim"]
    F2 = -(-(-(-Fdim // 2)) // 2)
    p = {}

    def glorot(shape, fan_in, fan_out):
        lim = math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * lim

    def trunc(shape, std=0.02):
        t = torch.empty(shape, dtype=torch.float64)
        torch.nn.init.trunc_normal_(t, 0.0, std, -2 * std, 2 * std, generator=g)
        return t

    nc = cfg.get("in_channels", 1)
    p["x_net.conv1.kernel"] = glorot((3, 3, nc, C), 9 * nc, 9 * C)
    p["x_net.conv2.kernel"] = glorot((3, 3, C, C), 9 * C, 9 * C)
    for bn in ("bn1", "bn2"):
        p["x_net.%s.gamma" % bn] = torch.ones(C, dtype=torch.float64)
        p["x_net.%s.beta" % bn] = torch.zeros(C, dtype=torch.float64)
        p["x_net.%s.moving_mean" % bn] = torch.zeros(C, dtype=torch.float64)
        p["x_net.%s.moving_variance" % bn] = torch.ones(C, dtype=torch.float64)
    p["x_net.dense.kernel"] = glorot((F2 * C, d), F2 * C, d)
    p["x_net.dense.bias"] = torch.zeros(d, dtype=torch.float64)
    p["y_net.embedding"] = (torch.rand((V, d), generator=g, dtype=torch.float64) - 0.5) * 0.1

    def mha(name):
        for w in ("wq", "wk", "wv", "dense"):
            p["%s.%s.kernel" % (name, w)] = trunc((d, d))
            p["%s.%s.bias" % (name, w)] = torch.zeros(d, dtype=torch.float64)

    def ffn_(name):
        p[name + ".0.kernel"] = trunc((d, dff))
        p[name + ".0.bias"] = torch.zeros(dff, dtype=torch.float64)
        p[name + ".1.kernel"] = trunc((dff, d))
        p[name + ".1.bias"] = torch.zeros(d, dtype=torch.float64)

    def ln(name):
        p[name + ".gamma"] = torch.ones(d, dtype=torch.float64)
        p[name + ".beta"] = torch.zeros(d, dtype=torch.float64)

    for i in range(cfg["num_encoder_layers"]):
        n = "encoder.%d" % i
        mha(n + ".self_attn"); ffn_(n + ".ffn"); ln(n + ".norm1"); ln(n + ".norm2")
    for i in range(cfg["num_decoder_layers"]):
        n = "decoder.%d" % i
        mha(n + ".attn1"); mha(n + ".attn2"); ffn_(n + ".ffn")
        ln(n + ".norm1"); ln(n + ".norm2"); ln(n + ".norm3")
    p["final_layer.kernel"] = glorot((d, V), d, V)
    p["final_layer.bias"] = torch.zeros(V, dtype=torch.float64)
    p["ctc_decoder.kernel"] = glorot((d, V), d, V)
    p["ctc_decoder.bias"] = torch.zeros(V, dtype=torch.float64)
    return {k: v.to(dtype) for k, v in p.items()}


def make_samples(B, T, Fdim, V, U, seed=1, full_length=True, dtype=torch.float32, in_channels=1):
    """Synthetic batch: N(0,1) fbank zeroed past input_length, labels uniform in [1, V-2],
    zero padded; label lengths in [U/2, U]."""
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, T, Fdim, in_channels)).astype(np.float32)
    if full_length:
        ilen = np.full((B,), T, dtype=np.int32)
    else:
        ilen = np.sort(rng.integers(int(0.6 * T), T + 1, size=B).astype(np.int32))[::-1].copy()
        ilen[0] = T
    for b in range(B):
        x[b, ilen[b]:] = 0.0
    olen = rng.integers(max(U // 2, 1), U + 1, size=B).astype(np.int32)
    olen[0] = U
    y = rng.integers(1, max(V - 1, 2), size=(B, U)).astype(np.int32)
    for b in range(B):
        y[b, olen[b]:] = 0
    return {
        "input": torch.from_numpy(x).to(dtype),
        "input_length": torch.from_numpy(ilen),
        "output": torch.from_numpy(y),
        "output_length": torch.from_numpy(olen),
    }
