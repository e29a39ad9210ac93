"""CTC and label-smoothed cross-entropy losses (mirror of the reference's athena/loss.py:24-80)."""
import torch

from . import ops
from .utils.misc import insert_eos_in_labels


def _as_padded_rows(logits):
    """Return (tensor whose storage the kernels read, ldl).  Logits produced by the model's
    vocabulary projections are views [..., :V] of rows padded to a multiple of 8 columns and are
    used in place; anything else is copied into such a buffer."""
    V = logits.shape[-1]
    try:
        _, _, ld = ops._rows2d(logits)
        ok = ld % 8 == 0 and logits.data_ptr() % 16 == 0 and logits.dtype in (torch.float32, torch.bfloat16)
    except RuntimeError:
        ok = False
    if ok:
        return logits, ld
    ld = ops.roundup(V, 8)
    buf = torch.zeros(logits.shape[:-1] + (ld,), dtype=torch.float32, device=logits.device)
    buf[..., :V].copy_(logits)
    return buf[..., :V], ld


class _CTCLossFn(torch.autograd.Function):
    """``upstream`` (a host float or None) is the factor the caller expects this loss to receive in the
    backward pass -- the weight of the CTC term in the total loss.  With it the forward pass leaves the
    gradient too, in the same pass over the logits (athena_ctc_loss_fwd_bwd), and the backward pass only
    corrects it on the device should the factor that arrives differ.  ``exact`` says that the caller
    is a model class of this package whose total loss carries this term with exactly that factor: only
    then, and only inside a solver step (whose root gradient is one), the comparison is left out.
    The gradient left by the forward pass is handed out once; a further backward pass over a retained
    graph recomputes it from the posteriors in the workspace."""

    @staticmethod
    def forward(ctx, logits, labels, label_len, logit_len, blank, bf16_grad=False, upstream=None, exact=False):
        B, T, V = logits.shape
        ctx.bf16_grad = bf16_grad
        ctx.exact = bool(exact)
        # no zero tensor for the gradient of the arg-max output (the engine would fill one per step)
        ctx.set_materialize_grads(False)
        U = labels.shape[1]
        src, ldl = _as_padded_rows(logits)
        labels = labels.to(torch.int32).contiguous()
        label_len = label_len.to(torch.int32).contiguous()
        logit_len = logit_len.to(torch.int32).contiguous()
        ws = torch.empty(ops.ctc_workspace_bytes(B, T, U), dtype=torch.uint8, device=logits.device)
        per = torch.empty(B, dtype=torch.float32, device=logits.device)
        # the pass over the rows also leaves the per-frame arg-max (zero beyond logit_length, written by the kernel)
        best = torch.empty(B * T, dtype=torch.int32, device=logits.device)
        ctx.grad = None
        if upstream and ctx.needs_input_grad[0]:
            gdt = torch.bfloat16 if bf16_grad else torch.float32
            ctx.grad = torch.empty((B, T, ldl), dtype=gdt, device=src.device)
            ctx.upstream = float(upstream)
            ops.ctc_loss_fwd_bwd(src, ldl, labels, label_len, logit_len, B, T, V, U, blank, per, ws,
                                 ctx.upstream / B, ctx.grad, argmax=best)
        else:
            ops.ctc_loss_fwd(src, ldl, labels, label_len, logit_len, B, T, V, U, blank, per, ws, argmax=best)
        out = torch.empty(1, dtype=torch.float32, device=logits.device)
        ops.group_sum(per, 1, B, 1.0 / B, out)          # tf.reduce_mean over the batch
        ctx.saved = (src, ldl, labels, label_len, logit_len, ws)
        ctx.dims = (B, T, V, U, blank)
        ctx.per_sample = per
        ctx.mark_non_differentiable(best)
        return out.view(()), best

    @staticmethod
    def backward(ctx, gout, _gbest=None):
        if gout is None:
            return None, None, None, None, None, None, None, None
        src, ldl, labels, label_len, logit_len, ws = ctx.saved
        B, T, V, U, blank = ctx.dims
        gs = gout.to(torch.float32).reshape(1).contiguous()
        if ctx.grad is not None:
            grad, ctx.grad = ctx.grad, None          # handed out once
            # the factor that arrives is compared on the device with the one announced and the gradient
            # corrected if they differ (a no-op launch when they agree)
            if not (ctx.exact and ops.root_gradient_is_one()):
                ops.rescale_unless(grad, gs, ctx.upstream)
        else:
            gdt = torch.bfloat16 if ctx.bf16_grad else torch.float32
            grad = torch.empty((B, T, ldl), dtype=gdt, device=src.device)
            ops.ctc_loss_bwd(src, ldl, labels, label_len, logit_len, B, T, V, U, blank, gs, 1.0 / B, grad, ws)
        if ctx.bf16_grad:
            return ops.bf16_grad_standin((B, T, V), grad), None, None, None, None, None, None, None
        return grad[..., :V], None, None, None, None, None, None, None


class _WeightedTotalFn(torch.autograd.Function):
    """total = sum_i weights[i] * terms[i] as one launch.  Inside a solver step (whose root gradient is
    one) the backward pass hands each term its weight as a constant that already sits on the device,
    so that no kernel runs at all."""

    @staticmethod
    def forward(ctx, weights, *terms):
        out = torch.empty(1, dtype=torch.float32, device=terms[0].device)
        ops.weighted_sum([t.detach().reshape(1) for t in terms], weights, out)
        ctx.weights = weights
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        grads = []
        for w in ctx.weights:
            c = ops.const_scalar(gout.device, w) if ops.root_gradient_is_one() else None
            grads.append(c if c is not None else gout * w)
        return (None,) + tuple(grads)


def weighted_total(terms, weights):
    """sum_i weights[i] * terms[i] for 0-dim fp32 loss tensors on one GPU (anything else: plain torch)."""
    if 0 < len(terms) <= 8 and all(isinstance(t, torch.Tensor) and t.is_cuda and t.dim() == 0 and
                                   t.dtype == torch.float32 for t in terms):
        return _WeightedTotalFn.apply(tuple(float(w) for w in weights), *terms)
    total = None
    for t, w in zip(terms, weights):
        total = t * w if total is None else total + t * w
    return total


class CTCLoss:
    """CTC loss: tf.nn.ctc_loss on batch-major logits with dense labels, then the batch mean
    (reference athena/loss.py:24-46)."""

    def __init__(self, logits_time_major=False, blank_index=-1, name="CTCLoss"):
        if logits_time_major:
            raise NotImplementedError("time-major logits are not built (the reference never sets it)")
        self.name = name
        self.logits_time_major = logits_time_major
        self.blank_index = blank_index
        self.need_logit_length = True
        self.takes_upstream = True

    def __call__(self, logits, samples, logit_length=None, upstream=None, exact=False):
        """``upstream``: the weight this loss will carry in the total loss, if the caller knows it (a
        host float); the gradient is then produced together with the loss.  ``exact``: the caller
        vouches that the total loss of a solver step carries this term with exactly that weight."""
        assert logit_length is not None
        V = logits.shape[-1]
        blank = self.blank_index + V if self.blank_index < 0 else self.blank_index
        loss, best = _CTCLossFn.apply(logits, samples["output"], samples["output_length"], logit_length, blank,
                                      ops.bf16_grad_wanted(logits), upstream, exact)
        # CTCAccuracy on the same logits and lengths picks this up instead of a second pass over them
        logits._athena_ctc_argmax = (best, logits._version, logit_length)
        return loss


class _CELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, ls, mode, bf16_grad=False):
        B, L, V = logits.shape
        ctx.bf16_grad = bf16_grad
        src, ldl = _as_padded_rows(logits)
        targets = targets.to(torch.int32).contiguous()
        rows = B * L
        row_loss = torch.empty(rows, dtype=torch.float32, device=logits.device)
        lse = torch.empty(rows, dtype=torch.float32, device=logits.device)
        ops.ce_loss_fwd(src, ldl, targets, rows, V, ls, row_loss, lse)
        out = torch.empty(1, dtype=torch.float32, device=logits.device)
        count = None
        if mode == "by_token":
            count = (targets != 0).sum().to(torch.float32).reshape(1)
            ops.group_sum(row_loss, 1, rows, 1.0, out)
            out = out / count
            scale = 1.0
        elif mode == "by_sequence":
            scale = 1.0 / B
            ops.group_sum(row_loss, 1, rows, scale, out)
        else:
            scale = 1.0 / rows
            ops.group_sum(row_loss, 1, rows, scale, out)
        ctx.saved = (src, ldl, targets, lse, count)
        ctx.dims = (B, L, V, ls, scale)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        src, ldl, targets, lse, count = ctx.saved
        B, L, V, ls, scale = ctx.dims
        gs = gout.to(torch.float32).reshape(1)
        if count is not None:
            gs = gs / count
        gs = gs.contiguous()
        gdt = torch.bfloat16 if ctx.bf16_grad else torch.float32
        grad = torch.empty((B, L, ldl), dtype=gdt, device=src.device)
        ops.ce_loss_bwd(src, ldl, targets, B * L, V, ls, lse, gs, scale, grad)
        if ctx.bf16_grad:
            return ops.bf16_grad_standin((B, L, V), grad), None, None, None, None
        return grad[..., :V], None, None, None, None


class Seq2SeqSparseCategoricalCrossentropy:
    """Cross-entropy at each target position of each sequence, with eos appended, padding
    (label 0) masked and optional label smoothing (reference athena/loss.py:49-80)."""

    def __init__(self, num_classes, eos=-1, by_token=False, by_sequence=True, from_logits=True,
                 label_smoothing=0.0):
        if not from_logits:
            raise NotImplementedError("only from_logits=True is built")
        self.by_token = by_token
        self.by_sequence = by_sequence
        self.num_classes = num_classes
        self.eos = num_classes + eos if eos < 0 else eos
        self.smoothing_rate = label_smoothing

    def __call__(self, logits, samples, logit_length=None):
        labels = insert_eos_in_labels(samples["output"], self.eos, samples["output_length"])
        seq_len = labels.shape[1]
        if seq_len != logits.shape[1]:
            logits = logits[:, :seq_len, :]
        mode = "by_token" if self.by_token else ("by_sequence" if self.by_sequence else "mean")
        return _CELossFn.apply(logits, labels, float(self.smoothing_rate), mode,
                               ops.bf16_grad_wanted(logits))


class _MPCLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, masked, logit_len):
        B, Tp, N = logits.shape
        src, ldl = _as_padded_rows(logits.to(torch.float32))
        target = target.to(torch.float32).contiguous()
        masked = masked.to(torch.float32).contiguous()
        logit_len = logit_len.to(torch.int32).contiguous()
        row_loss = torch.empty(B * Tp, dtype=torch.float32, device=logits.device)
        ops.mpc_loss_fwd(src, ldl, target, masked, logit_len, B, Tp, N, row_loss)
        out = torch.empty(1, dtype=torch.float32, device=logits.device)
        ops.group_sum(row_loss, 1, B * Tp, 1.0 / (B * Tp), out)
        ctx.saved = (src, ldl, target, masked, logit_len)
        ctx.dims = (B, Tp, N)
        return out.view(())

    @staticmethod
    def backward(ctx, gout):
        src, ldl, target, masked, logit_len = ctx.saved
        B, Tp, N = ctx.dims
        gs = gout.to(torch.float32).reshape(1).contiguous()
        grad = torch.empty((B, Tp, ldl), dtype=torch.float32, device=src.device)
        ops.mpc_loss_bwd(src, ldl, target, masked, logit_len, B, Tp, N, gs, 1.0 / (B * Tp), grad)
        return grad[..., :N], None, None, None


class MPCLoss:
    """L1 loss on the masked acoustic features of a batch (reference athena/loss.py:83-105):
    |target - logits| where the model's input was masked to 0 and the frame is inside the utterance,
    summed over the feature axis, averaged over all (batch, frame) positions."""

    def __init__(self, name="MPCLoss"):
        self.name = name

    def __call__(self, logits, samples, logit_length=None):
        B, Tp, N = logits.shape
        target, masked = samples["output"], samples["input"]
        if target.numel() != B * Tp * N or masked.numel() != B * Tp * N:
            raise ValueError("MPCLoss: output %s / input %s do not reshape to the logits %s" %
                             (tuple(target.shape), tuple(masked.shape), tuple(logits.shape)))
        return _MPCLossFn.apply(logits, target, masked, logit_length)
