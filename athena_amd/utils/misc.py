"""Label / mask helpers (mirror of the reference's athena/utils/misc.py:32-126,209-217) on torch tensors.

These are tiny integer index manipulations on [B, U] label tensors (plumbing around the kernels);
they run on whatever device the labels live on.
"""
import torch

from ..layers.attention import AttnMask


def insert_sos_in_labels(labels, sos):
    """[sos, labels...] (misc.py:32-34)."""
    return torch.cat([torch.full_like(labels[:, :1], sos), labels], dim=-1)


class StepLabels:
    """Labels, lengths and key masks of the batch a model is working on, produced by one kernel
    (``ops.prepare_labels``) when the model's forward pass starts.  ``insert_eos_in_labels`` and
    ``compute_logit_length`` hand out these tensors when they are asked about the same batch (same
    tensor objects, unchanged since), instead of launching their own little kernels."""
    last = None

    def __init__(self, labels, label_len, input_len, eos, y0, tgt, y_mask, logit_len, x_mask):
        self.src = (labels, label_len, input_len)
        self.versions = tuple(t._version for t in self.src)
        self.eos = eos
        self.y0, self.tgt, self.y_mask, self.logit_len, self.x_mask = y0, tgt, y_mask, logit_len, x_mask

    def fresh(self):
        return all(t._version == v for t, v in zip(self.src, self.versions))

    @classmethod
    def build(cls, samples, Tp, sos, eos):
        """None when the batch does not have the form the kernel takes (then the callers fall back
        to the tensor operations below)."""
        from .. import ops
        labels, label_len, input_len = samples["output"], samples["output_length"], samples["input_length"]
        ok = all(isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.int32 and t.is_contiguous()
                 for t in (labels, label_len, input_len)) and labels.dim() == 2 and labels.shape[1] > 0
        if not ok:
            cls.last = None
            return None
        cls.last = cls(labels, label_len, input_len, eos, *ops.prepare_labels(labels, label_len, input_len, Tp, sos, eos))
        return cls.last

    @classmethod
    def targets_for(cls, labels, eos, labels_length):
        sl = cls.last
        if sl is not None and sl.src[0] is labels and sl.src[1] is labels_length and sl.eos == eos and sl.fresh():
            return sl.tgt
        return None

    @classmethod
    def logit_length_for(cls, input_length):
        sl = cls.last
        if sl is not None and sl.src[2] is input_length and sl.fresh():
            return sl.logit_len
        return None


def insert_eos_in_labels(input_labels, eos, labels_length):
    """labels ++ [0] with eos written at position labels_length (misc.py:50-57)."""
    hit = StepLabels.targets_for(input_labels, eos, labels_length)
    if hit is not None:
        return hit
    zero = torch.zeros_like(input_labels[:, :1])
    labels = torch.cat([input_labels, zero], dim=-1)
    idx = labels_length.to(torch.int64).view(-1, 1)
    add = torch.zeros_like(labels).scatter_(1, idx, torch.full_like(idx, eos, dtype=labels.dtype))
    return labels + add


def remove_eos_in_labels(input_labels, labels_length):
    """misc.py:37-47"""
    labels = input_labels[:, :-1]
    length = labels_length - 1
    ar = torch.arange(labels.shape[1], device=labels.device)
    return labels * (ar[None, :] < length[:, None]).to(labels.dtype)


def generate_square_subsequent_mask(size, device=None):
    """float mask, 1.0 above the diagonal (misc.py:60-65)."""
    return 1 - torch.tril(torch.ones((size, size), device=device))


def create_multihead_mask(x, x_length, y, reverse=False):
    """Masks for multi-head attention (misc.py:68-90) as ``AttnMask`` objects: the encoder/memory
    mask hides frames >= x_length; the decoder mask hides keys whose label is 0 and keys ahead of
    the query."""
    if reverse:
        raise NotImplementedError("create_multihead_mask(reverse=True) is not built")
    x_mask, y_mask = None, None
    if x is not None:
        T = x.shape[1]
        ar = torch.arange(T, device=x_length.device)
        x_mask = AttnMask((ar[None, :] >= x_length[:, None]).to(torch.uint8).contiguous(), False)
    if y is not None:
        y_mask = AttnMask((y == 0).to(torch.uint8).contiguous(), True)
    return x_mask, y_mask


def validate_seqs(seqs, eos):
    """Per row: tokens before the first ``eos`` with zeros dropped (misc.py:98-126, where the
    result is a SparseTensor).  Returns (list of python lists, number of kept tokens)."""
    out, count = [], 0
    for row in seqs.tolist():
        keep = []
        if eos != 0:
            for c in row:
                if c == eos:
                    break
                keep.append(c)
        else:
            keep = list(row)
        keep = [c for c in keep if c != 0]
        out.append(keep)
        count += len(keep)
    return out, float(count)


def apply_label_smoothing(inputs, num_classes, smoothing_rate=0.1):
    """misc.py:209-217"""
    return (1.0 - smoothing_rate) * inputs + smoothing_rate / num_classes
