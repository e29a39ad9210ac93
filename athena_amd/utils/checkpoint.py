"""Checkpoint manager on torch state (mirror of the reference's athena/utils/checkpoint.py:24-120).

Same files in the checkpoint directory as the reference leaves there, so scripts that look at them
keep working: ``ckpt-N`` snapshots (here one ``torch.save`` file each instead of a TensorFlow
index/data pair), ``checkpoint`` and ``best_loss`` holding ``model_checkpoint_path: "ckpt-N"``, and
``n_best`` with one ``name<TAB>metric`` line per snapshot.
"""
import logging
import math
import os
import re

import torch


def _latest(directory, latest_filename="checkpoint"):
    """Path named by a ``model_checkpoint_path: "..."`` file (tf.train.latest_checkpoint), or None."""
    path = os.path.join(directory, latest_filename)
    if not os.path.exists(path):
        return None
    with open(path) as f:
        m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', f.read())
    if not m:
        return None
    full = os.path.join(directory, m.group(1))
    return full if os.path.exists(full) else None


class Checkpoint:
    """``Checkpoint(checkpoint_directory, model=model, optimizer=optimizer)``; calling it saves a
    snapshot and records best loss / per-snapshot metric (checkpoint.py:45-75,110-113).  Restores
    the latest snapshot of the directory on construction, like the reference."""

    def __init__(self, checkpoint_directory=None, model=None, **kwargs):
        self.model = model
        self.tracked = dict(kwargs)                       # e.g. optimizer=...
        self.best_loss = math.inf
        self.n_best_model = {}
        self.save_counter = 0
        self.metric_name = model.metric.name if getattr(model, "metric", None) is not None else None
        if checkpoint_directory is None:
            checkpoint_directory = os.path.join(os.path.expanduser("~"), ".athena")
        self.checkpoint_prefix = os.path.join(checkpoint_directory, "ckpt")
        self.checkpoint_directory = checkpoint_directory
        logging.info("trying to restore from : %s", checkpoint_directory)
        self.restore(_latest(checkpoint_directory))
        n_best = os.path.join(checkpoint_directory, "n_best")
        if os.path.exists(n_best):
            with open(n_best) as f:
                for line in f:
                    key, val = line.split("\t")
                    self.n_best_model[key] = float(val.strip())

    # ---- save / restore ------------------------------------------------------------------------
    def save(self, file_prefix=None):
        prefix = file_prefix or self.checkpoint_prefix
        os.makedirs(os.path.dirname(prefix), exist_ok=True)
        self.save_counter += 1
        path = "%s-%d" % (prefix, self.save_counter)
        state = {"save_counter": self.save_counter, "model": self.model.store.state_dict()}
        for name, obj in self.tracked.items():
            if hasattr(obj, "state_dict"):
                state[name] = obj.state_dict()
        torch.save(state, path)
        with open(os.path.join(os.path.dirname(prefix), "checkpoint"), "w") as wf:
            wf.write('model_checkpoint_path: "%s"' % os.path.basename(path))
        return path

    def restore(self, path):
        """Load a snapshot (no-op for None, as tf.train.Checkpoint.restore(None))."""
        if path is None:
            return
        state = torch.load(path, map_location="cpu", weights_only=True)
        self.model.store.load_state_dict(state["model"])
        for name, obj in self.tracked.items():
            if name in state and hasattr(obj, "load_state_dict"):
                obj.load_state_dict(state[name])
        self.save_counter = int(state.get("save_counter", self.save_counter))

    def restore_from_best(self):
        """checkpoint.py:115-122"""
        self.restore(_latest(self.checkpoint_directory, latest_filename="best_loss"))

    # ---- bookkeeping ---------------------------------------------------------------------------
    def _compare_and_save_best(self, loss, metrics, save_path):
        """checkpoint.py:60-75"""
        checkpoint = save_path.split("/")[-1]
        if loss is not None and float(loss) < self.best_loss:
            self.best_loss = float(loss)
            with open(os.path.join(self.checkpoint_directory, "best_loss"), "w") as wf:
                wf.write('model_checkpoint_path: "%s"' % checkpoint)
        if metrics is None or len(metrics) == 0 or self.metric_name is None:
            return
        self.n_best_model[checkpoint] = float(metrics[self.metric_name])
        with open(os.path.join(self.checkpoint_directory, "n_best"), "w") as wf:
            for key in self.n_best_model:
                wf.write("%s\t%s\n" % (key, float(self.n_best_model[key])))

    def compute_nbest_avg(self, model_avg_num):
        """Average the trainable variables of the ``model_avg_num`` snapshots with the highest
        metric into the live model (checkpoint.py:77-108)."""
        avg_file = os.path.join(self.checkpoint_directory, "n_best")
        if not os.path.exists(avg_file):
            self.restore_from_best()
            return
        scores = {}
        with open(avg_file) as f:
            for line in f:
                key, val = line.split("\t")
                scores[key] = float(val.strip())
        ckpt_list = [k for k, _ in sorted(scores.items(), key=lambda item: item[1], reverse=True)[:model_avg_num]]
        logging.info("n_best_metrics_checkpoint: %s", ckpt_list)
        store = self.model.store
        total = None
        for key in ckpt_list:
            self.restore(os.path.join(self.checkpoint_directory, key))      # current variables are updated
            cur = store.params[:store.n_train].detach().clone()
            total = cur if total is None else total + cur
        if total is not None:
            store.params[:store.n_train].copy_(total / len(ckpt_list))
            store.shadows_dirty = True

    def __call__(self, loss=None, metrics=None):
        logging.info("saving model in :%s", self.checkpoint_prefix)
        save_path = self.save(file_prefix=self.checkpoint_prefix)
        self._compare_and_save_best(loss, metrics, save_path)
