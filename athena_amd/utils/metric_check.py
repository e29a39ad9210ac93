"""Log-line formatter for training / evaluation (mirror of the reference's
athena/utils/metric_check.py:22-120; the tf.summary scalars are replaced by an optional
tab-separated ``scalars.tsv`` in the summary directory)."""
import os
import time

_SUMMARY_DIR = None


def set_default_summary_writer(summary_directory=None):
    """reference athena/utils/misc.py:188-193"""
    global _SUMMARY_DIR
    if summary_directory is None:
        summary_directory = os.path.join(os.path.expanduser("~"), ".athena", "event")
    os.makedirs(summary_directory, exist_ok=True)
    _SUMMARY_DIR = summary_directory


def _scalar(tag, value, step):
    if _SUMMARY_DIR is None:
        return
    with open(os.path.join(_SUMMARY_DIR, "scalars.tsv"), "a") as f:
        if hasattr(value, "detach"):
            value = value.detach()
        f.write("%d\t%s\t%.8g\n" % (step, tag, float(value)))


class MetricChecker:
    """Formats ``global_steps / learning_rate / loss / metrics / sec/iter`` lines."""

    def __init__(self, optimizer):
        self.optimizer = optimizer
        self.time_last_call = time.time()
        self.steps_last_call = 0

    def __call__(self, loss, metrics, evaluate_epoch=-1):
        if evaluate_epoch == -1:
            return self.summary_train(loss, metrics)
        return self.summary_evaluate(loss, metrics, evaluate_epoch)

    def summary_train(self, loss, metrics):
        global_steps = int(self.optimizer.iterations)
        learning_rate = self.optimizer.lr(global_steps) if callable(self.optimizer.lr) else self.optimizer.lr
        total_loss = sum(list(loss.values())) if isinstance(loss, dict) else loss
        _scalar("total_loss", total_loss, global_steps)
        if isinstance(loss, dict):
            for name in loss:
                _scalar(name, loss[name], global_steps)
        _scalar("learning_rate", learning_rate, global_steps)
        reports = ""
        reports += "global_steps: %d\t" % (global_steps)
        reports += "learning_rate: %.4e\t" % (learning_rate)
        reports += "loss: %.4f\t" % (float(total_loss.detach() if hasattr(total_loss, "detach") else total_loss))
        if isinstance(loss, dict):
            for name in loss:
                reports += "%s: %.4f\t" % (name, float(loss[name]))
        if metrics is not None:
            for name in metrics:
                metric = float(metrics[name])
                _scalar(name, metric, global_steps)
                reports += "%s: %.4f\t" % (name, metric)
        right_now = time.time()
        duration = right_now - self.time_last_call
        self.time_last_call = right_now
        if self.steps_last_call != 0:
            sec_per_iter = duration / float(global_steps - self.steps_last_call)
            reports += "sec/iter: %.4f" % (sec_per_iter)
        self.steps_last_call = global_steps
        return reports

    def summary_evaluate(self, loss, metrics, epoch=-1):
        reports = ""
        global_steps = int(self.optimizer.iterations)
        total_loss = sum(list(loss.values())) if isinstance(loss, dict) else loss
        if epoch >= 0:
            _scalar("evaluate_total_loss", total_loss, global_steps)
            reports += "epoch: %d\t" % (epoch)
        reports += "loss: %.4f\t" % (float(total_loss.detach() if hasattr(total_loss, "detach") else total_loss))
        if isinstance(loss, dict):
            for name in loss:
                reports += "%s: %.4f\t" % (name, float(loss[name]))
        if metrics is not None:
            for name in metrics:
                metric = float(metrics[name])
                if epoch >= 0:
                    _scalar("evaluate_" + name, metric, global_steps)
                reports += "%s: %.4f\t" % (name, metric)
        return reports
