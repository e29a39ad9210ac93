"""TEST INFRASTRUCTURE -- CPU restatement of the reference's one-utterance decoding path: beam search
over the attention decoder (athena/tools/beam_search.py:243-343) with the CTC prefix scorer
(athena/tools/ctc_scorer.py:39-173), on top of the model restated in oracle/model.py
(SpeechTransformer.decode / time_propagate, athena/models/speech_transformer.py:164-238;
MtlTransformerCtc.decode, athena/models/mtl_seq2seq.py:117-142).

Parity unpinned: TensorFlow cannot be imported here and the reference keeps no decoding fixtures; the
prefix scorer below is checked against an enumeration of all CTC paths in tests/test_decode_cpu.py.

Written hypothesis by hypothesis and label by label with plain Python loops (float32 scores as in
the reference); only tests may import it.
"""
import math

import numpy as np
import torch

from . import model as om

LOGZERO = -10000000000.0


def _lae(a, b):
    return np.logaddexp(np.float32(a), np.float32(b)).astype(np.float32)


def prefix_scores(x, blank, eos, y, cs, r_prev):
    """CTCPrefixScorer.cand_score (ctc_scorer.py:122-173) for ONE hypothesis: x [T, V] log-probs,
    y the hypothesis with <sos>, cs the candidate labels, r_prev [T, 2] its state.
    Returns ([len(cs)] scores, [len(cs), T, 2] states)."""
    T = x.shape[0]
    out_len = len(y) - 1
    scores, states = [], []
    r_sum = [_lae(r_prev[t][0], r_prev[t][1]) for t in range(T)]
    for c in cs:
        r = [[np.float32(LOGZERO), np.float32(LOGZERO)] for _ in range(T)]
        if out_len == 0:
            r[0][0] = np.float32(x[0][c])
        phi = [np.float32(r_prev[t][1]) if (out_len > 0 and c == y[-1]) else r_sum[t] for t in range(T)]
        start = max(out_len, 1)
        psi = r[start - 1][0]
        for t in range(start, T):
            r[t][0] = np.float32(_lae(r[t - 1][0], phi[t - 1]) + np.float32(x[t][c]))
            r[t][1] = np.float32(_lae(r[t - 1][0], r[t - 1][1]) + np.float32(x[t][blank]))
            psi = _lae(psi, np.float32(phi[t - 1] + np.float32(x[t][c])))
        if c == eos:
            psi = r_sum[T - 1]
        scores.append(np.float32(psi))
        states.append(r)
    return scores, states


def decoder_step(p, cfg, seqs, memory, memory_mask):
    """SpeechTransformer.time_propagate: logits [n, V] of the newest position of every hypothesis."""
    n = len(seqs)
    y0 = torch.tensor(seqs, dtype=torch.long)
    L = y0.shape[1]
    y = om.y_net(p, y0, cfg, False)
    look_ahead = (1 - torch.tril(torch.ones(L, L, dtype=y.dtype)))[None, None]
    mem = memory.expand(n, -1, -1)
    mm = memory_mask.expand(n, -1, -1, -1)
    out = y
    for i in range(cfg["num_decoder_layers"]):
        out, _ = om.decoder_layer(p, "decoder.%d" % i, out, mem, look_ahead, mm, cfg, False)
    return om.dense(out, p, "final_layer")[:, -1, :]


def encode(p, samples, cfg):
    """Encoder output [1, T', d] and its float mask [1, 1, 1, T'] in inference mode."""
    x = om.x_net(p, samples["input"], cfg, False)
    length = om.compute_logit_length(samples["input_length"])
    mask = (torch.arange(x.shape[1])[None, :] >= length[:, None]).to(x.dtype)[:, None, None, :]
    mem = x
    for i in range(cfg["num_encoder_layers"]):
        mem = om.encoder_layer(p, "encoder.%d" % i, mem, mask, cfg, False)
    return mem, mask


def beam_search(p, samples, cfg, beam_size, ctc_weight=0.0):
    """The decoded label ids of one utterance (without <sos>, with the final <eos>)."""
    V = cfg["num_class"]
    sos = eos = blank = V - 1
    with torch.no_grad():
        memory, mask = encode(p, samples, cfg)
        x = None
        if ctc_weight != 0:
            x = torch.log_softmax(om.dense(memory, p, "ctc_decoder").float(), dim=-1)[0].numpy().astype(np.float32)
    T = memory.shape[1]
    ctc_beam = beam_size * 2
    init_state = None
    if x is not None:
        init_state = [[np.float32(LOGZERO), np.float32(0)] for _ in range(T)]
        acc = np.float32(0)
        for t in range(T):
            acc = np.float32(acc + x[t][blank])
            init_state[t][1] = acc
    # a hypothesis: (labels with sos, score, {label: (ctc state, ctc total)})
    hyps = [([sos], np.float32(0), {})]
    finished = []            # (score, labels, length)
    for pos in range(T):
        with torch.no_grad():
            logits = decoder_step(p, cfg, [h[0] for h in hyps], memory, mask)
            logp = torch.log_softmax(logits.float(), dim=-1).numpy().astype(np.float32)
        rows, tables = [], []
        for i, (seq, score, table) in enumerate(hyps):
            row = (logp[i] + score).astype(np.float32)
            new_table = {}
            if x is not None:
                state, total = table.get(seq[-1], (init_state, np.float32(0)))
                top = sorted(np.argsort(row)[-ctc_beam:].tolist())
                psi, states = prefix_scores(x, blank, eos, seq, top, state)
                extra = np.full(V, 0.0, dtype=np.float32)
                for k, c in enumerate(top):
                    extra[c] = np.float32(np.float32(ctc_weight) * np.float32(psi[k] - total) + np.float32(500))
                    new_table[c] = (states[k], psi[k])
                extra = (extra - np.float32(500)).astype(np.float32)
                row = (row + extra).astype(np.float32)
            rows.append(row)
            tables.append(new_table)
        for i, (seq, _, _) in enumerate(hyps):
            finished.append((rows[i][eos], seq + [eos] * (T - len(seq)), len(seq) + 1))
        if len(finished) > beam_size:
            order = sorted(range(len(finished)), key=lambda j: (-np.float32(finished[j][0] / np.float32(finished[j][2])), j))
            finished = [finished[j] for j in order[:beam_size]]
        if pos + 1 >= T:
            break
        flat = []
        for i in range(len(hyps)):
            for c in range(V):
                if c != eos:
                    flat.append((rows[i][c], i, c))
        best_new = max(f[0] for f in flat)
        new_len = np.float32(len(hyps[0][0]) + 1)
        if best_new < min(f[0] for f in finished) and \
                np.float32(best_new / new_len) < min(np.float32(f[0] / np.float32(f[2])) for f in finished):
            break
        order = sorted(range(len(flat)), key=lambda j: (-flat[j][0], j))[:beam_size]
        hyps = [(hyps[flat[j][1]][0] + [flat[j][2]], flat[j][0], tables[flat[j][1]]) for j in order]
    best = max(range(len(finished)), key=lambda j: (np.float32(finished[j][0] / np.float32(finished[j][2])), -j))
    score, labels, length = finished[best]
    return labels[1:length]
