"""Masked predictive coding: the pre-training sibling of the speech transformer (mirror of the
reference's athena/models/masked_pc.py:29-213).  Same x_net and encoder stack as ``SpeechTransformer``
(so ``restore_from_pretrained_model(mpc_model, "mpc")`` can carry the weights over), a Dense layer
back to four frames of acoustic features per encoder position, and an L1 loss on the frames that
were masked out of the input."""
import torch

from ..layers.commons import Dense, bind_all
from ..layers.transformer import TransformerEncoder, TransformerEncoderLayer
from ..loss import MPCLoss
from ..runtime import ParamStore
from ..utils.hparam import register_and_parse_hparams
from ..metrics import _Deferred
from .base import BaseModel, LazyMetric
from .speech_transformer import _PRECISIONS, SpeechTransformer, XNet


class _MeanMetric:
    """tf.keras.metrics.Mean of the loss values it is given.  The values stay on the device; they
    are brought to the host and added up when the metric is read (or flushed), so that updating
    it costs no synchronisation and can be recorded in a HIP graph."""

    def __init__(self, name):
        self.name = name
        self._pending = []
        self.reset_states()

    def reset_states(self):
        self.flush()
        self.total, self.count = 0.0, 0

    def _add(self, value):
        self.total += float(value.reshape(-1)[0])
        self.count += 1

    def update_state(self, value):
        self._pending.append(_Deferred(self, [value.detach().to(torch.float32).reshape(1)], self._add))

    def flush(self):
        pending, self._pending = self._pending, []
        for item in pending:
            item()

    def result(self):
        self.flush()
        return self.total / self.count if self.count else 0.0


class MaskedPredictCoding(BaseModel):
    """x_net -> encoder -> Dense(4 * feature size); trained to reproduce the masked frames."""
    default_config = {
        "return_encoder_output": False,
        "num_filters": 512,
        "d_model": 512,
        "num_heads": 8,
        "num_encoder_layers": 12,
        "dff": 1280,
        "rate": 0.1,
        "chunk_size": 3,
        "keep_probability": 0.85,
        "input_dropout_rate": 0.0
    }

    def __init__(self, data_descriptions, config=None, precision=None, store=None, seed=1, device=None):
        super().__init__()
        self.downsample_scale = 4
        self.num_class = data_descriptions.num_class * self.downsample_scale
        self.hparams = register_and_parse_hparams(self.default_config, config, cls=self.__class__)
        shape = data_descriptions.sample_shape["input"]
        self.dim, self.num_channels = int(shape[-2]), int(shape[-1])
        self.loss_function = MPCLoss()
        self.metric = _MeanMetric(name="AverageLoss")
        if self.hparams.input_dropout_rate:
            raise NotImplementedError("input_dropout_rate > 0 is not built")
        own_store = store is None
        if own_store:
            import os
            precision = precision or os.environ.get("ATHENA_PRECISION", "bf16")
            store = ParamStore(_PRECISIONS[precision], device=device)
        self.store = store
        hp = self.hparams
        self.x_net = XNet(store, shape, hp.num_filters, hp.d_model, hp.rate)
        self.encoder = TransformerEncoder([
            TransformerEncoderLayer(store, "encoder.%d" % i, hp.d_model, hp.num_heads, hp.dff, hp.rate, "gelu")
            for i in range(hp.num_encoder_layers)])
        self.final_layer = Dense(store, "final_layer", hp.d_model, self.num_class)
        self.mask_generator = torch.Generator().manual_seed(seed)
        if own_store:
            self.store.finalize(seed)
            bind_all(self)

    def reset_metrics(self):
        self.metric.reset_states()

    def call(self, samples, training=None):
        self.store.refresh_shadows()
        x = self.x_net(samples["input"], training=training)
        x = self.encoder(x, None, training=training)
        return SpeechTransformer._project(self.final_layer, x)

    def get_loss(self, logits, samples, training=None):
        logit_length = self.compute_logit_length(samples)
        loss = self.loss_function(logits, samples, logit_length)
        self.metric.update_state(loss)
        return loss, {self.metric.name: LazyMetric(self.metric)}

    def compute_logit_length(self, samples):
        n = samples["input_length"].to(torch.int32)
        return (((n + 1) // 2) + 1) // 2

    def generate_mpc_mask(self, input_data):
        """[batch, seq_len, dim, channels] of 1.0 (keep) / 0.0 (masked): whole chunks of
        ``chunk_size * 4`` frames are dropped with probability 1 - keep_probability
        (masked_pc.py:160-191)."""
        batch, seq_len, dim, num_channels = input_data.shape
        chunk_size = self.hparams.chunk_size * self.downsample_scale
        if (1 - self.hparams.keep_probability) * seq_len <= chunk_size:
            chunk_size = self.downsample_scale
        num_chunk = -(-seq_len // chunk_size)
        keep = (torch.rand((batch, num_chunk), generator=self.mask_generator) <
                self.hparams.keep_probability).to(input_data.dtype)
        mask = keep[:, :, None].expand(batch, num_chunk, chunk_size).reshape(batch, -1)[:, :seq_len]
        return mask[:, :, None, None].expand(batch, seq_len, dim, num_channels).to(input_data.device)

    def prepare_samples(self, samples):
        """Cut the features to a multiple of four frames; ``input`` = masked features, ``output`` =
        the features themselves as [batch, seq_len, dim * channels], lengths capped to the cut
        (masked_pc.py:193-213)."""
        data = samples["input"]
        seq_len = data.shape[1] // self.downsample_scale * self.downsample_scale
        data = data[:, :seq_len]
        data = torch.as_tensor(data).to(device=self.store.device, dtype=torch.float32)
        length = torch.clamp(torch.as_tensor(samples["input_length"]).to(device=self.store.device, dtype=torch.int32),
                             max=seq_len)
        out = {k: v for k, v in samples.items() if k not in ("input", "output", "input_length", "output_length")}
        out["input"] = (data * self.generate_mpc_mask(data)).contiguous()
        out["output"] = data.reshape(data.shape[0], seq_len, -1).contiguous()
        out["input_length"] = length
        out["output_length"] = length
        return out
