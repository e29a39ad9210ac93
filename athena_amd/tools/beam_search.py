"""Beam search over the attention decoder with optional CTC prefix scores (reference
athena/tools/beam_search.py:29-343).

The search runs on the host over small numpy arrays -- a beam of at most ``beam_size`` hypotheses of
one utterance; the only device work is ``decoder_one_step`` (the model's ``time_propagate``: y_net,
decoder stack and output layer on the hypotheses so far).  Scores are float32 as in the reference;
ties between equal scores go to the lower flat index, like tf.math.top_k.
"""
import collections

import numpy as np
import torch

from .ctc_scorer import CTCPrefixScorer

CandidateHolder = collections.namedtuple(
    "CandidateHolder", ["cand_seqs", "cand_logits", "cand_states", "cand_scores", "cand_parents"])


def _top(values, k):
    """Indices of the k largest entries, best first, the lower index first among equals."""
    order = np.argsort(-values, kind="stable")
    return order[:k]


def _select(state, parents):
    """A per-hypothesis state after the beam was re-ordered by ``parents``."""
    if state is None:
        return None
    if hasattr(state, "reorder"):
        return state.reorder(parents)
    if isinstance(state, torch.Tensor):
        return state[torch.as_tensor(np.asarray(parents), device=state.device)]
    return np.asarray(state)[np.asarray(parents)]


class _Finished:
    """The pool of hypotheses that have taken <eos>: at most ``keep`` of them, ranked by score per
    token (reference beam_search.py:151-194)."""

    def __init__(self, keep, width, eos):
        self.keep, self.width, self.eos = keep, width, eos
        self.scores = np.zeros(0, dtype=np.float32)
        self.seqs = np.zeros((0, width), dtype=np.int64)
        self.lengths = np.zeros(0, dtype=np.int64)

    def add(self, seqs, scores):
        n, length = seqs.shape
        padded = np.full((n, self.width), self.eos, dtype=np.int64)
        padded[:, :length] = seqs
        self.scores = np.concatenate([self.scores, scores.astype(np.float32)])
        self.seqs = np.concatenate([self.seqs, padded])
        self.lengths = np.concatenate([self.lengths, np.full(n, length + 1, dtype=np.int64)])
        if len(self.scores) > self.keep:
            best = _top(self.per_token(), self.keep)
            self.scores, self.seqs, self.lengths = self.scores[best], self.seqs[best], self.lengths[best]

    def per_token(self):
        return self.scores / self.lengths.astype(np.float32)

    def best(self):
        i = _top(self.per_token(), 1)[0]
        return self.seqs[i, 1:self.lengths[i]]


class BeamSearchDecoder:
    """Beam search decoding used in seq2seq decoder layer."""

    def __init__(self, num_class, sos, eos, beam_size):
        self.num_class = num_class
        self.sos = sos
        self.eos = eos
        self.beam_size = beam_size
        self.scorers = []
        self.ctc_scorer = None
        self.lm_model = None
        self.states = []
        self.decoder_one_step = None

    @staticmethod
    def build_decoder(hparams, num_class, sos, eos, decoder_one_step, lm_model=None):
        """The decoder for ``hparams`` (decoder_type, beam_size, ctc_weight, lm_weight) around the
        model's time-propagating function (beam_search.py:54-101)."""
        beam = hparams.decoder_type == "beam_search_decoder"
        dec = BeamSearchDecoder(num_class, sos, eos, beam_size=hparams.beam_size if beam else 1)
        dec.decoder_one_step = decoder_one_step
        if beam and hparams.ctc_weight != 0:
            dec.set_ctc_scorer(CTCPrefixScorer(eos, ctc_beam=hparams.beam_size * 2, num_classes=num_class,
                                               ctc_weight=hparams.ctc_weight))
        # a language-model scorer is attached only for lm_type "ngram" / "rnn" (beam_search.py:78-94); any
        # other lm_type -- the default "" included -- decodes without one whatever lm_weight says
        lm_type = getattr(hparams, "lm_type", "")
        if getattr(hparams, "lm_weight", 0) != 0 and lm_type in ("ngram", "rnn"):
            if lm_type == "ngram" and getattr(hparams, "lm_path", None) is None:
                raise ValueError("lm path should not be none")
            raise NotImplementedError("language-model scorers (n-gram, RNN) are not built")
        return dec

    def set_lm_model(self, lm_model):
        self.lm_model = lm_model
        self.scorers.append(lm_model)

    def set_ctc_scorer(self, ctc_scorer):
        self.ctc_scorer = ctc_scorer
        self.scorers.append(ctc_scorer)

    def beam_search_score(self, candidate_holder, encoder_outputs):
        """One decoder step on every live hypothesis: scores [n, num_class] of extending it by each
        class (beam_search.py:120-150)."""
        logits, history, states = self.decoder_one_step(candidate_holder.cand_logits, candidate_holder.cand_seqs,
                                                        self.states, encoder_outputs)
        self.states = states
        logprobs = torch.log_softmax(logits.to(torch.float32), dim=-1).cpu().numpy().astype(np.float32)
        new_scores = logprobs + candidate_holder.cand_scores[:, None].astype(np.float32)
        new_states = candidate_holder.cand_states
        for scorer in self.scorers:
            other, new_states = scorer.score(candidate_holder, new_scores)
            if other is not None:
                new_scores = new_scores + other
        return new_scores, history, new_states

    def __call__(self, cand_seqs, cand_states, init_states, encoder_outputs):
        """``cand_seqs``: the <sos> start [1, 1]; ``cand_states``: [cand_seqs, scorer states ...];
        ``init_states``: the decoder step counter; ``encoder_outputs``: (encoder output
        [1, T', d], its mask).  Returns the best hypothesis [1, length] (int64, without <sos>, ending
        in <eos>) (beam_search.py:243-343)."""
        seqs = np.asarray(torch.as_tensor(cand_seqs).cpu()).reshape(1, -1).astype(np.int64)
        holder = CandidateHolder(seqs, [], list(cand_states), np.zeros(1, dtype=np.float32), np.zeros(1, dtype=np.int64))
        holder.cand_states[0] = seqs
        self.states = init_states
        if self.lm_model is not None:
            self.lm_model.reset()
        enc, enc_mask = encoder_outputs
        max_seq_len = enc.shape[1]
        done = _Finished(self.beam_size, max_seq_len, self.eos)
        open_classes = np.array([c for c in range(self.num_class) if c != self.eos])
        for pos in range(max_seq_len):
            n = holder.cand_seqs.shape[0]
            tiled = (enc[:1].expand(n, -1, -1), _tile_mask(enc_mask, n))
            new_scores, history, new_states = self.beam_search_score(holder, tiled)
            done.add(holder.cand_seqs, new_scores[:, self.eos])
            if pos + 1 >= max_seq_len:
                break
            live = new_scores[:, open_classes]
            length = np.float32(holder.cand_seqs.shape[1] + 1)
            # every continuation is already worse than every finished hypothesis, in total and per token
            if live.max() < done.scores.min() and (live / length).max() < done.per_token().min():
                break
            flat = live.reshape(-1)
            pick = _top(flat, self.beam_size) if flat.size > self.beam_size else np.arange(flat.size)
            parents, syms = pick // len(open_classes), open_classes[pick % len(open_classes)]
            seqs = np.concatenate([holder.cand_seqs[parents], syms[:, None]], axis=1)
            states = [_select(s, parents) for s in new_states]
            states[0] = seqs
            holder = CandidateHolder(seqs, [_select(h, parents) for h in history], states, flat[pick], parents)
        return torch.from_numpy(done.best()[None].astype(np.int64))


def _tile_mask(mask, n):
    if mask is None:
        return None
    if hasattr(mask, "key_mask"):                        # AttnMask
        km = mask.key_mask
        return type(mask)(None if km is None else km[:1].expand(n, -1).contiguous(), mask.causal)
    return mask[:1].expand(n, *mask.shape[1:])
