"""Oracle (test infrastructure): CTC loss and its gradient on the CPU.

The reference computes CTC with ``tf.nn.ctc_loss(labels=dense int32, logits,
label_length, logit_length, logits_time_major=False, blank_index=-1)``
followed by ``tf.reduce_mean`` (reference ``athena/loss.py:35-46``).  With a
dense ``labels`` tensor TensorFlow 2.0/2.1 (pinned by the reference's
``README.md:69`` / ``docker/Dockerfile:1``; not vendored, not installed here)
takes its pure-TF ``ctc_loss_dense`` branch.  Two restatements live here:

* :func:`ctc_loss_tf_dense` follows that branch's published algorithm step by
  step, including its quirks (blank rotated to class 0, a silent start state,
  finite ``log(1e-307)`` instead of ``-inf`` for the initial/final state
  vectors, a forward scan that is not length-masked, the loss read from the
  backward scan).  It builds dense ``[B, S, S]`` transition matrices, so it is
  for small cases.
* :func:`ctc_loss_textbook` is the classic alpha/beta recursion on the
  ``2U+1`` blank-augmented lattice in float64 with a true ``-inf``; it is the
  banded, fast form used at benchmark sizes.

Both return per-sample losses and ``d loss_b / d logits`` (no batch mean).
:func:`ctc_brute_force` enumerates every alignment and is the known-answer
generator.  PARITY UNPINNED against TensorFlow itself (see package docstring).
"""
import itertools

import numpy as np


def _logsumexp(a, axis=-1, keepdims=False):
    m = np.max(a, axis=axis, keepdims=True)
    m_safe = np.where(np.isfinite(m), m, 0.0)
    with np.errstate(divide="ignore"):
        out = np.log(np.sum(np.exp(a - m_safe), axis=axis, keepdims=True)) + m_safe
    if not keepdims:
        out = np.squeeze(out, axis=axis)
    return out


def log_softmax(x, axis=-1):
    return x - _logsumexp(x, axis=axis, keepdims=True)


# --------------------------------------------------------------------------
# TensorFlow ``ctc_loss_dense`` restated (reference call site athena/loss.py:38-45)
# --------------------------------------------------------------------------
def ctc_loss_tf_dense(logits, labels, label_length, logit_length, blank_index=-1,
                      dtype=np.float64):
    """Per-sample loss [B] and gradient [B,T,V] the way TF's dense branch computes them."""
    logits = np.asarray(logits, dtype=dtype)
    labels = np.asarray(labels, dtype=np.int64)
    label_length = np.asarray(label_length, dtype=np.int64)
    logit_length = np.asarray(logit_length, dtype=np.int64)
    B, T, V = logits.shape
    U = labels.shape[1]
    if blank_index < 0:
        blank_index += V
    # blank moved to class 0, labels below the blank shifted up by one
    if blank_index != 0:
        logits_r = np.concatenate(
            [logits[:, :, blank_index:blank_index + 1], logits[:, :, :blank_index],
             logits[:, :, blank_index + 1:]], axis=2)
        labels_r = np.where(labels < blank_index, labels + 1, labels)
    else:
        logits_r, labels_r = logits, labels
    # time major
    lp = log_softmax(np.transpose(logits_r, (1, 0, 2)))            # [T,B,V]
    log0 = dtype(np.float32(np.log(np.float64(0) + 1e-307)))       # finite "log zero", cast as TF does

    n_lab = U + 1                     # label states 0..U, state 0 is the silent start state
    S = 2 * n_lab                     # blank states n_lab..2*n_lab-1
    # emission log-probs per state: [start(-inf), label_1..label_U, blank_0..blank_U]
    lab_lp = np.take_along_axis(lp, np.broadcast_to(labels_r[None], (T, B, U)), axis=2)
    blank_lp = np.repeat(lp[:, :, :1], n_lab, axis=2)
    obs = np.concatenate([np.full((T, B, 1), -np.inf, dtype=dtype), lab_lp, blank_lp], axis=2)

    # transition matrix trans[b, to, from]
    trans = np.zeros((B, S, S), dtype=dtype)
    trans[:, np.arange(S), np.arange(S)] = 1.0                     # self loops
    trans[:, 1, 0] += 1.0                                          # start -> first label
    for i in range(1, n_lab):
        trans[:, i, n_lab + i - 1] += 1.0                          # blank_{i-1} -> label_i
    for i in range(n_lab):
        trans[:, n_lab + i, i] += 1.0                              # label_i (or start) -> blank_i
    for i in range(2, n_lab):                                      # label_{i-1} -> label_i unless repeated
        rep = labels_r[:, i - 2] == labels_r[:, i - 1]
        trans[:, i, i - 1] += 1.0 - rep.astype(dtype)
    with np.errstate(divide="ignore"):
        ltrans = np.log(trans)
    ltrans_bwd = np.transpose(ltrans, (0, 2, 1))

    init = np.full((B, S), log0, dtype=dtype)
    init[:, 0] = 0.0
    final = np.full((B, S), log0, dtype=dtype)
    final[np.arange(B), label_length] = 0.0
    final[np.arange(B), label_length + n_lab] = 0.0

    mask = (np.arange(T)[:, None] < logit_length[None, :]).astype(dtype)   # [T,B]

    # forward scan (inclusive, over ALL T steps, renormalised each step)
    fwd = np.empty((T + 1, B, S), dtype=dtype)
    fwd[0] = init
    for t in range(T):
        s = _logsumexp(fwd[t][:, None, :] + ltrans, axis=-1) + obs[t]
        s = s - _logsumexp(s, axis=-1, keepdims=True)
        fwd[t + 1] = s
    # backward scan (reverse, length-masked)
    bwd = np.empty((T + 1, B, S), dtype=dtype)
    bwd[T] = final
    cum = np.zeros((B,), dtype=dtype)
    for t in range(T - 1, -1, -1):
        s = bwd[t + 1] + obs[t]
        s = _logsumexp(s[:, None, :] + ltrans_bwd, axis=-1)
        z = _logsumexp(s, axis=-1, keepdims=True)
        s = s - z
        cum = cum + z[:, 0] * mask[t]
        m = mask[t][:, None]
        # masked steps pass the final-state vector through
        bwd[t] = np.where(m > 0, s, final)
    fb = fwd[1:] + bwd[1:]
    fb = fb - _logsumexp(fb, axis=2, keepdims=True)
    with np.errstate(divide="ignore"):
        fb = fb + np.log(mask)[:, :, None]
    log_likelihood = bwd[0][:, 0] + cum

    # state posteriors -> class posteriors (rotated class order)
    occ = np.zeros((T, B, V), dtype=dtype)
    post = np.exp(fb)
    for b in range(B):
        np.add.at(occ[:, b, :], (slice(None), labels_r[b]), post[:, b, 1:n_lab])
    occ[:, :, 0] += post[:, :, n_lab:].sum(axis=2)
    grad_r = (np.exp(lp) - occ) * mask[:, :, None]
    # undo the blank rotation
    if blank_index != 0:
        grad = np.concatenate(
            [grad_r[:, :, 1:blank_index + 1], grad_r[:, :, :1], grad_r[:, :, blank_index + 1:]], axis=2)
    else:
        grad = grad_r
    return -log_likelihood, np.transpose(grad, (1, 0, 2))


# --------------------------------------------------------------------------
# Classic alpha/beta on the 2U+1 lattice (float64, true -inf)
# --------------------------------------------------------------------------
def ctc_loss_textbook(logits, labels, label_length, logit_length, blank_index=-1):
    """Per-sample loss [B] and gradient (softmax - occupancy)*mask [B,T,V], float64."""
    logits = np.asarray(logits, dtype=np.float64)
    labels = np.asarray(labels, dtype=np.int64)
    label_length = np.asarray(label_length, dtype=np.int64)
    logit_length = np.asarray(logit_length, dtype=np.int64)
    B, T, V = logits.shape
    U = labels.shape[1]
    blank = blank_index + V if blank_index < 0 else blank_index
    S = 2 * U + 1
    lp = log_softmax(logits)                                       # [B,T,V]
    ext = np.full((B, S), blank, dtype=np.int64)
    ext[:, 1::2] = labels
    em = np.take_along_axis(lp, np.broadcast_to(ext[:, None, :], (B, T, S)), axis=2)   # [B,T,S]
    # skip transition s-2 -> s allowed for label states whose label differs from the previous label
    skip = np.zeros((B, S), dtype=bool)
    if U > 1:
        skip[:, 3::2] = labels[:, 1:] != labels[:, :-1]
    s_idx = np.arange(S)[None, :]
    valid = s_idx < (2 * label_length[:, None] + 1)                # states that exist for this sample
    NEG = -np.inf

    def shift(a, k):
        out = np.full_like(a, NEG)
        out[:, k:] = a[:, :-k] if k > 0 else a
        return out

    alpha = np.full((B, T, S), NEG)
    a0 = np.full((B, S), NEG)
    a0[:, 0] = em[:, 0, 0]
    if S > 1:
        a0[:, 1] = np.where(label_length > 0, em[:, 0, 1], NEG)
    alpha[:, 0] = np.where(valid, a0, NEG)
    for t in range(1, T):
        prev = alpha[:, t - 1]
        cand = np.stack([prev, shift(prev, 1), np.where(skip, shift(prev, 2), NEG)], axis=0)
        a = _logsumexp(cand, axis=0) + em[:, t]
        alpha[:, t] = np.where(valid, a, NEG)
    # beta[b,t,s]: log prob of emitting frames t+1.. from state s at t and ending in a final state at len-1
    beta = np.full((B, T, S), NEG)
    last = logit_length - 1
    fin = np.zeros((B, S), dtype=bool)
    fin[np.arange(B), 2 * label_length] = True
    fin[np.arange(B), np.maximum(2 * label_length - 1, 0)] |= label_length > 0
    bnext = np.full((B, S), NEG)
    skip_from = np.zeros((B, S), dtype=bool)                       # s -> s+2 allowed
    skip_from[:, :-2] = skip[:, 2:]

    def shiftl(a, k):
        out = np.full_like(a, NEG)
        out[:, :-k] = a[:, k:]
        return out

    for t in range(T - 1, -1, -1):
        is_last = (last == t)[:, None]
        active = (t < logit_length)[:, None]
        nxt = bnext + (em[:, t + 1] if t + 1 < T else 0.0)         # beta[t+1] + emission at t+1
        cand = np.stack([nxt, shiftl(nxt, 1), np.where(skip_from, shiftl(nxt, 2), NEG)], axis=0)
        b = _logsumexp(cand, axis=0)
        b = np.where(is_last, np.where(fin, 0.0, NEG), b)
        b = np.where(active & valid, b, NEG)
        beta[:, t] = b
        bnext = b
    ll0 = _logsumexp(alpha[:, 0] + beta[:, 0], axis=-1)
    ll = np.where(logit_length > 0, ll0, NEG)
    with np.errstate(invalid="ignore"):
        post = np.exp(alpha + beta - ll[:, None, None])            # [B,T,S]
    post = np.where(np.isfinite(post), post, 0.0)
    occ = np.zeros((B, T, V))
    for b in range(B):
        np.add.at(occ[b], (slice(None), ext[b]), post[b])
    mask = (np.arange(T)[None, :] < logit_length[:, None]).astype(np.float64)
    grad = (np.exp(lp) - occ) * mask[:, :, None]
    return -ll, grad


# --------------------------------------------------------------------------
# Brute force over all alignments (known-answer generator, tiny cases only)
# --------------------------------------------------------------------------
def collapse(path, blank):
    out, prev = [], None
    for p in path:
        if p != prev and p != blank:
            out.append(p)
        prev = p
    return out


def ctc_brute_force(logits_tv, label_seq, blank):
    """-log sum over all length-T alignments that collapse to ``label_seq``."""
    logits_tv = np.asarray(logits_tv, dtype=np.float64)
    T, V = logits_tv.shape
    lp = log_softmax(logits_tv)
    target = [int(x) for x in label_seq]
    terms = []
    for path in itertools.product(range(V), repeat=T):
        if collapse(path, blank) == target:
            terms.append(sum(lp[t, p] for t, p in enumerate(path)))
    if not terms:
        return np.inf
    return -float(_logsumexp(np.array(terms)))


def ctc_mean_loss(logits, labels, label_length, logit_length, blank_index=-1):
    """``tf.reduce_mean`` of the per-sample losses (reference athena/loss.py:46) and its gradient."""
    loss, grad = ctc_loss_textbook(logits, labels, label_length, logit_length, blank_index)
    B = loss.shape[0]
    return loss.mean(), grad / B
