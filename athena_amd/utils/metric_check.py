"""Log lines for training / evaluation.

The line format is the contract with the reference (athena/utils/metric_check.py:51-120):
tab-separated ``name: value`` fields -- ``global_steps``, ``learning_rate``, ``loss``, one field per
named loss and per metric, and ``sec/iter`` from the second report on (training);
``epoch`` (when one is given), ``loss`` and the metrics (evaluation).  In place of the reference's
``tf.summary.scalar`` events every reported scalar is appended to ``scalars.tsv`` in the summary
directory (``step <tab> tag <tab> value``).
"""
import os
import time


class _ScalarLog:
    """Append-only ``scalars.tsv`` in the summary directory; inactive until a directory is set."""
    directory = None

    @classmethod
    def open(cls, directory):
        os.makedirs(directory, exist_ok=True)
        cls.directory = directory

    @classmethod
    def write(cls, step, rows):
        if cls.directory is None or not rows:
            return
        with open(os.path.join(cls.directory, "scalars.tsv"), "a") as f:
            f.writelines("%d\t%s\t%.8g\n" % (step, tag, val) for tag, val in rows)


def set_default_summary_writer(summary_directory=None):
    """Where the scalars go; default ~/.athena/event (reference athena/utils/misc.py:188-193)."""
    _ScalarLog.open(summary_directory or os.path.join(os.path.expanduser("~"), ".athena", "event"))


def _num(x):
    return float(x.detach()) if hasattr(x, "detach") else float(x)


def _loss_fields(loss):
    """[("loss", total)] followed by the named parts of a dict-valued loss."""
    if isinstance(loss, dict):
        parts = [(name, _num(v)) for name, v in loss.items()]
        return [("loss", sum(v for _, v in parts))] + parts
    return [("loss", _num(loss))]


class MetricChecker:
    """Callable that turns (loss, metrics) into the report line and logs the scalars.

    ``checker(loss, metrics)`` reports a training step; ``checker(loss, metrics, epoch)`` an
    evaluation (``epoch`` < -1: a running report inside an evaluation, without epoch and scalars)."""

    def __init__(self, optimizer):
        self.optimizer = optimizer
        self._clock = time.time()
        self._steps_at_clock = 0

    def __call__(self, loss, metrics, evaluate_epoch=-1):
        if evaluate_epoch == -1:
            return self.summary_train(loss, metrics)
        return self.summary_evaluate(loss, metrics, evaluate_epoch)

    def _step_and_rate(self):
        step = int(self.optimizer.iterations)
        lr = self.optimizer.lr
        return step, float(lr(step) if callable(lr) else lr)

    def summary_train(self, loss, metrics):
        step, lr = self._step_and_rate()
        losses = _loss_fields(loss)
        scores = [(name, float(v)) for name, v in (metrics or {}).items()]
        _ScalarLog.write(step, [("total_loss", losses[0][1])] + losses[1:] + [("learning_rate", lr)] + scores)
        line = ["global_steps: %d" % step, "learning_rate: %.4e" % lr]
        line += ["%s: %.4f" % kv for kv in losses + scores]
        now = time.time()
        elapsed, self._clock = now - self._clock, now
        tail = ""
        if self._steps_at_clock:
            tail = "sec/iter: %.4f" % (elapsed / float(step - self._steps_at_clock))
        self._steps_at_clock = step
        return "\t".join(line) + "\t" + tail

    def summary_evaluate(self, loss, metrics, epoch=-1):
        step, _ = self._step_and_rate()
        losses = _loss_fields(loss)
        scores = [(name, float(v)) for name, v in (metrics or {}).items()]
        line = []
        if epoch >= 0:
            _ScalarLog.write(step, [("evaluate_total_loss", losses[0][1])] +
                             [("evaluate_" + name, v) for name, v in losses[1:] + scores])
            line.append("epoch: %d" % epoch)
        line += ["%s: %.4f" % kv for kv in losses + scores]
        return "\t".join(line) + "\t"
