"""Base class for models (mirror of the reference's athena/models/base.py:24-74)."""
import logging

import torch
from torch import nn


class BaseModel(nn.Module):
    """Base class for model."""

    def __init__(self, **kwargs):
        super().__init__()
        self.loss_function = None
        self.metric = None

    def call(self, samples, training=None):
        raise NotImplementedError()

    def forward(self, samples, training=None):
        return self.call(samples, training=training)

    def get_loss(self, outputs, samples, training=None):
        """loss of ``loss_function`` and the running metric (base.py:37-49)."""
        logit_length = None
        if self.loss_function is None:
            loss = 0.0
        else:
            logit_length = self.compute_logit_length(samples)
            if getattr(self.loss_function, "takes_upstream", False):
                # this loss is the whole objective: its gradient arrives with factor 1
                loss = self.loss_function(outputs, samples, logit_length, upstream=1.0, exact=True)
            else:
                loss = self.loss_function(outputs, samples, logit_length)
        if self.metric is None:
            metrics = {}
        else:
            self.metric(outputs, samples, logit_length)
            metrics = {self.metric.name: LazyMetric(self.metric)}
        return loss, metrics

    def compute_logit_length(self, samples):
        return samples["input_length"]

    def reset_metrics(self):
        if self.metric is not None:
            self.metric.reset_states()

    def prepare_samples(self, samples):
        """Move a batch to the model's device with the dtypes the kernels expect
        (shapes unchanged, as in the reference)."""
        return samples

    def restore_from_pretrained_model(self, pretrained_model, model_type=""):
        logging.info("restore from pretrained model")

    def decode(self, samples, hparams, decoder):
        logging.info("sorry, this model do not support decode")


class LazyMetric:
    """A metric value that is computed (host-side scoring, one device synchronisation) only when
    it is read: ``float(m)``, formatting with ``%``, or ``m.value``."""

    def __init__(self, metric):
        self._metric = metric

    @property
    def value(self):
        return self._metric.result()

    def __float__(self):
        return float(self._metric.result())

    def __repr__(self):
        return "%.4f" % float(self)

    def __format__(self, spec):
        return format(float(self), spec)


def to_device_samples(samples, device):
    """samples dict -> device tensors: input fp32, lengths/labels int32."""
    out = {}
    for k, v in samples.items():
        t = torch.as_tensor(v)
        if k == "input":
            t = t.to(device=device, dtype=torch.float32)
        elif k in ("input_length", "output", "output_length"):
            t = t.to(device=device, dtype=torch.int32)
        else:
            t = t.to(device)
        out[k] = t
    return out
