"""CTC prefix scores for one-pass hybrid CTC/attention decoding (reference
athena/tools/ctc_scorer.py:21-173; Watanabe et al., "Hybrid CTC/attention architecture for
end-to-end speech recognition").

For a hypothesis h (labels after <sos>) and a next label c the prefix score is
log sum_{t} phi_{t-1}(h) p_t(c): the probability of every label sequence that starts with h.c, where
phi_{t-1}(h) is the probability of having emitted exactly h up to frame t-1 (not ending in c when c
repeats the last label of h).  The recursion over the frames keeps, per hypothesis and label, the
pair (r_t^n, r_t^b) = "h.c up to frame t ending in a non-blank / in a blank".

Everything is float32 numpy on the host, as in the reference.  All live hypotheses of a beam have the
same length, so one call advances the whole beam: the arrays below carry a hypothesis axis and a
candidate axis, and only the loop over the frames is sequential.
"""
import numpy as np
import torch


class PrefixTable:
    """What the scorer remembers of each live hypothesis: the labels it scored in the last step
    (``labels`` [n, k], sorted), their recursion states (``states`` [n, k, T, 2]) and their total
    prefix scores (``totals`` [n, k]).  A label that was not among the k scored ones has the
    initial state and total 0, which is what the reference's dense [n, num_classes, ...] tables
    hold for it.  ``select`` follows the beam's parent pointers."""

    def __init__(self, labels, states, totals, init_state):
        self.labels, self.states, self.totals, self.init_state = labels, states, totals, init_state

    def reorder(self, parents):
        parents = np.asarray(parents)
        return PrefixTable(self.labels[parents], self.states[parents], self.totals[parents], self.init_state)

    def lookup(self, symbols):
        """(state [n, T, 2], total [n]) of the label each hypothesis was extended with."""
        n = len(symbols)
        states = np.repeat(self.init_state[None], n, axis=0)
        totals = np.zeros(n, dtype=np.float32)
        for i, s in enumerate(symbols):
            hit = np.nonzero(self.labels[i] == s)[0]
            if len(hit):
                states[i] = self.states[i, hit[0]]
                totals[i] = self.totals[i, hit[0]]
        return states, totals


class CTCPrefixScorer:
    """ctc one pass decoding scorer; ``score`` shares the interface of the other scorers of
    ``BeamSearchDecoder`` (candidate_holder, new_scores) -> (scores to add, cand_states)."""

    def __init__(self, eos, ctc_beam, num_classes, blank=-1, ctc_weight=0.25):
        self.logzero = -10000000000.0
        self.eos = eos
        self.blank = blank % num_classes
        self.ctc_beam = ctc_beam
        self.state_index = 0
        self.score_index = 0
        self.ctc_weight = ctc_weight
        self.num_classes = num_classes
        self.x = None
        self.input_length = None
        self.init_state = None

    def initial_state(self, init_cand_states, x):
        """``x``: log-softmax of the CTC logits, [beam, T, num_classes] (the first entry is used).
        Appends the scorer's table to ``init_cand_states`` (and a place holder where the reference
        keeps its second table, so the indices of later states agree)."""
        x0 = x[0]
        self.x = (x0.detach().cpu().numpy() if isinstance(x0, torch.Tensor) else np.asarray(x0)).astype(np.float32)
        self.input_length = len(self.x)
        r = np.full((self.input_length, 2), self.logzero, dtype=np.float32)
        r[:, 1] = np.cumsum(self.x[:, self.blank], dtype=np.float32)
        self.init_state = r
        empty = PrefixTable(np.zeros((1, 0), dtype=np.int64), np.zeros((1, 0, self.input_length, 2), dtype=np.float32),
                            np.zeros((1, 0), dtype=np.float32), r)
        init_cand_states.append(empty)
        self.state_index = len(init_cand_states) - 1
        init_cand_states.append(None)
        self.score_index = len(init_cand_states) - 1
        return init_cand_states

    def score(self, candidate_holder, new_scores):
        seqs = np.asarray(candidate_holder.cand_seqs)
        cand_states = candidate_holder.cand_states
        table = cand_states[self.state_index]
        new_scores = np.asarray(new_scores)
        n = seqs.shape[0]
        prev_state, prev_total = table.lookup(seqs[:, -1])
        # the ctc_beam best labels of each hypothesis by the scores so far, in label order
        k = min(self.ctc_beam, self.num_classes)
        labels = np.sort(np.argsort(new_scores, axis=1)[:, -k:], axis=1)
        psi, states = self.cand_score(seqs, labels, prev_state)
        result = np.full((n, self.num_classes), -500.0, dtype=np.float32)
        delta = (np.float32(self.ctc_weight) * (psi - prev_total[:, None]) + np.float32(500.0)) - np.float32(500.0)
        np.put_along_axis(result, labels, delta.astype(np.float32), axis=1)
        cand_states[self.state_index] = PrefixTable(labels, states, psi, self.init_state)
        return result, cand_states

    def cand_score(self, y, cs, r_prev):
        """Prefix scores of ``y[i]`` extended by each of ``cs[i]`` and the new recursion states.
        y [n, L] (with <sos>), cs [n, k], r_prev [n, T, 2] -> (log_psi [n, k], r [n, k, T, 2])."""
        y = np.asarray(y)
        if y.ndim == 1:                     # one hypothesis, the reference's call form
            psi, r = self.cand_score(y[None], np.asarray(cs)[None], np.asarray(r_prev)[None])
            return psi[0], r[0]
        n, k = cs.shape
        T = self.input_length
        out_len = y.shape[1] - 1            # labels after <sos>
        xs = self.x[:, cs]                  # [T, n, k]
        xb = self.x[:, self.blank]
        r = np.full((T, 2, n, k), self.logzero, dtype=np.float32)
        if out_len == 0:
            r[0, 0] = xs[0]
        r_sum = np.logaddexp(r_prev[:, :, 0], r_prev[:, :, 1]).astype(np.float32)      # [n, T]
        log_phi = np.repeat(r_sum.T[:, :, None], k, axis=2)                            # [T, n, k]
        if out_len > 0:
            repeat = cs == y[:, -1:]                                                    # [n, k]
            log_phi = np.where(repeat[None], r_prev[:, :, 1].T[:, :, None], log_phi)
        start = max(out_len, 1)
        log_psi = r[start - 1, 0].copy()
        for t in range(start, T):
            r[t, 0] = np.logaddexp(r[t - 1, 0], log_phi[t - 1]) + xs[t]
            r[t, 1] = np.logaddexp(r[t - 1, 0], r[t - 1, 1]) + xb[t]
            log_psi = np.logaddexp(log_psi, log_phi[t - 1] + xs[t])
        # ending the hypothesis here: the probability of exactly y
        log_psi = np.where(cs == self.eos, r_sum[:, -1:], log_psi).astype(np.float32)
        return log_psi, np.ascontiguousarray(np.transpose(r, (2, 3, 0, 1)))
