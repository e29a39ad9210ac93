"""Accuracy metrics of the training step (mirror of the reference's athena/metrics.py:33-132).

Everything of an update runs on the GPU inside the step: the arg-max over the vocabulary
(``athena_argmax_rows``, or the CTC loss kernel's by-product), then the greedy CTC collapse / the cut
at eos and the Levenshtein distance against the non-zero labels (``athena_seq_errors``, one
workgroup per utterance), which adds (errors, labels) to a two-element device accumulator.  The host
reads those two integers when the metric is read (every ``log_interval`` steps), so a training step
enqueues no device-to-host synchronisation and no host-side scoring.  Hypotheses that already are
Python lists (beam-search output) are scored on the host.
"""
import logging

import torch

from . import ops
from .utils.misc import validate_seqs


def edit_distance(a, b):
    """Levenshtein distance between two sequences (tf.edit_distance with normalize=False)."""
    la, lb = len(a), len(b)
    if la == 0:
        return lb
    if lb == 0:
        return la
    prev = list(range(lb + 1))
    for i in range(1, la + 1):
        cur = [i] + [0] * lb
        ai = a[i - 1]
        for j in range(1, lb + 1):
            cur[j] = min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ai != b[j - 1]))
        prev = cur
    return prev[lb]


class _Sum:
    """tf.keras.metrics.Sum"""

    def __init__(self):
        self.value = 0.0

    def __call__(self, v):
        self.value += float(v)

    def result(self):
        return self.value

    def reset_states(self):
        self.value = 0.0


class Accuracy:
    """Base class: accuracy = 1 - errors / total (metrics.py:33-63)."""

    def __init__(self, name="Accuracy", rank_size=1):
        self.name = name
        self.error_count = _Sum()
        self.total_count = _Sum()
        self.rank_size = rank_size
        self._pending = []
        self._dev_counts = None      # int64 [2] on the device: (errors, labels) since the last reset

    def reset_states(self):
        self._drain()
        self.error_count.reset_states()
        self.total_count.reset_states()
        if self._dev_counts is not None:
            self._dev_counts.zero_()

    def device_counts(self, device):
        """The device accumulator the scoring kernel adds to.  It is created by the first update,
        which is never a recorded one (a recorded step keeps adding to this same buffer)."""
        if self._dev_counts is None or self._dev_counts.device != device:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the first update of a metric must not happen inside a graph capture")
            self._dev_counts = torch.zeros(2, dtype=torch.int64, device=device)
        return self._dev_counts

    def update_state(self, predictions, samples, logit_length=None):
        raise NotImplementedError

    def __call__(self, logits, samples, logit_length=None):
        return self.update_state(logits, samples, logit_length)

    def _drain(self):
        pending, self._pending = self._pending, []
        for fn in pending:
            fn()

    def flush(self):
        """Move queued device results to pinned host memory (asynchronously) so the device
        buffers may be reused; the host-side scoring still waits for ``result()``."""
        new = []
        for item in self._pending:
            new.append(item.to_host() if hasattr(item, "to_host") else item)
        self._pending = new

    def result(self):
        self._drain()
        err, tot = self.error_count.result(), self.total_count.result()
        if self._dev_counts is not None:
            dev = self._dev_counts.tolist()      # the one device-to-host read of the metric
            err, tot = err + float(dev[0]), tot + float(dev[1])
        if tot == 0.0:
            return 0.0 if err > 0 else float("nan")   # x/0 = inf -> 0.0 ; 0/0 = nan
        return 1.0 - err / tot


class _Deferred:
    """Device tensors of one update, scored on the host later."""

    def __init__(self, owner, tensors, score):
        self.owner, self.tensors, self.score, self.event = owner, tensors, score, None

    def to_host(self):
        if self.event is None and any(t.is_cuda for t in self.tensors):
            # all of them in one launch (the pinned buffers are addressable from the device)
            host, pairs = [], []
            for t in self.tensors:
                h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                if t.is_cuda and t.is_contiguous():
                    pairs.append((t, h))
                else:
                    h.copy_(t, non_blocking=True)
                host.append(h)
            if pairs:
                ops.copy_many(pairs)
            self.event = torch.cuda.Event()
            self.event.record()
            self.tensors = host
        return self

    def __call__(self):
        self.to_host()
        if self.event is not None:
            self.event.synchronize()
        self.score(*[t.cpu() for t in self.tensors])


class CharactorAccuracy(Accuracy):
    """Character accuracy from edit distances against the non-zero labels (metrics.py:66-89)."""

    def __init__(self, name="CharactorAccuracy", rank_size=1):
        super().__init__(name=name, rank_size=rank_size)

    def update_state(self, predictions, samples, logit_length=None):
        """``predictions``: list of python int lists (one hypothesis per utterance)."""
        labels = samples["output"].cpu().tolist() if isinstance(samples["output"], torch.Tensor) else samples["output"]
        return self._score(predictions, labels)

    def _score(self, hyps, labels):
        errs, count = 0.0, 0.0
        for hyp, row in zip(hyps, labels):
            ref = [c for c in row if c != 0]
            errs += edit_distance(hyp, ref)
            count += len(ref)
        if self.rank_size > 1:
            import torch.distributed as dist
            t = torch.tensor([errs, count], dtype=torch.float64)
            if dist.is_initialized():
                dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else "cpu"
                t = t.to(dev)
                dist.all_reduce(t)
                errs, count = float(t[0]), float(t[1])
        self.error_count(errs)
        self.total_count(count)
        return errs, count


class Seq2SeqSparseCategoricalAccuracy(CharactorAccuracy):
    """Attention-decoder accuracy: argmax, cut at eos, drop zeros (metrics.py:92-106)."""

    def __init__(self, eos, name="Seq2SeqSparseCategoricalAccuracy"):
        super().__init__(name=name)
        self.eos = eos

    def __call__(self, logits, samples, logit_length=None):
        B, L, V = logits.shape
        _, _, ld = ops._rows2d(logits)
        pred = torch.empty((B, L), dtype=torch.int32, device=logits.device)
        ops.argmax_rows(logits.detach(), ld, B * L, V, pred)
        ops.seq_errors(pred, None, _labels_i32(samples["output"], logits.device), 1, int(self.eos),
                       counts=self.device_counts(logits.device))


def _labels_i32(labels, device):
    """The label matrix as the scoring kernel reads it (the batches of ``prepare_samples`` already are)."""
    t = torch.as_tensor(labels)
    if t.device != device or t.dtype != torch.int32 or not t.is_contiguous():
        t = t.to(device=device, dtype=torch.int32).contiguous()
    return t


class CTCAccuracy(CharactorAccuracy):
    """Greedy CTC decoding accuracy (metrics.py:109-132): argmax per frame up to logit_length,
    merge repeats, drop the blank (last class)."""

    def __init__(self, name="CTCAccuracy"):
        super().__init__(name=name)
        self.need_logit_length = True

    def __call__(self, logits, samples, logit_length=None):
        assert logit_length is not None
        B, T, V = logits.shape
        if T <= samples["output"].shape[1] + 1:
            logging.warning("the length of logits is shorter than that of labels")
            return
        hit = getattr(logits, "_athena_ctc_argmax", None)
        if hit is not None and hit[1] == logits._version and hit[2] is logit_length and hit[0].numel() == B * T:
            pred = hit[0]        # left by the CTC loss kernel for the frames inside logit_length
        else:
            _, _, ld = ops._rows2d(logits)
            pred = torch.empty(B * T, dtype=torch.int32, device=logits.device)
            ops.argmax_rows(logits.detach(), ld, B * T, V, pred)

        lens = logit_length if (logit_length.dtype == torch.int32 and logit_length.is_contiguous()) \
            else logit_length.to(torch.int32).contiguous()
        ops.seq_errors(pred.view(B, T), lens, _labels_i32(samples["output"], logits.device), 0, V - 1,
                       counts=self.device_counts(logits.device))
