"""Speech transformer (mirror of the reference's athena/models/speech_transformer.py:34-162)."""
import logging

import torch
from torch import nn

from .. import ops
from ..layers.commons import BatchNormParams, Conv3x3Params, Dense, bind_all
from ..layers.fused import DenseFn, XNetFn, YNetFn
from ..layers.transformer import Transformer
from ..loss import Seq2SeqSparseCategoricalCrossentropy
from ..metrics import Seq2SeqSparseCategoricalAccuracy
from ..runtime import ParamStore, uniform
from ..utils.hparam import register_and_parse_hparams
from ..layers.attention import AttnMask
from ..utils.misc import StepLabels, create_multihead_mask, insert_sos_in_labels
from .base import BaseModel, to_device_samples

_PRECISIONS = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16, "fp32": torch.float32, "float32": torch.float32}


class XNet(nn.Module):
    """Conv2D(3x3,s2,same,no bias)-BN-relu6 twice, reshape to (T', F'*C), Dense(d)+relu6,
    positional encoding, dropout (speech_transformer.py:72-105)."""

    def __init__(self, store, in_shape, num_filters, d_model, rate):
        super().__init__()
        self.store, self.num_filters, self.d_model, self.rate = store, num_filters, d_model, rate
        feat, nc = int(in_shape[-2]), int(in_shape[-1])
        if nc > 4:
            raise NotImplementedError("x_net: more than 4 input channels is not built")
        f2 = ops.same_out(ops.same_out(feat))
        self.feat_dim, self.in_channels = feat, nc
        self.conv1 = Conv3x3Params(store, "x_net.conv1", nc, num_filters, as_matrix=False)
        self.bn1 = BatchNormParams(store, "x_net.bn1", num_filters)
        self.conv2 = Conv3x3Params(store, "x_net.conv2", num_filters, num_filters, as_matrix=True)
        self.bn2 = BatchNormParams(store, "x_net.bn2", num_filters)
        self.dense = Dense(store, "x_net.dense", f2 * num_filters, d_model, tiled=d_model == 256)   # tiled copies: input gradient as one panel launch
        self.site_drop = store.new_site()

    def forward(self, x0, training=None):
        if x0.shape[-2] != self.feat_dim or x0.shape[-1] != self.in_channels:
            raise ValueError("x_net: input feature shape %s, expected (%d, %d)" %
                             (tuple(x0.shape[-2:]), self.feat_dim, self.in_channels))
        return XNetFn.apply(x0, self, bool(training), *self.parameters())


class YNet(nn.Module):
    """Embedding(V, d) (uniform +-0.05), scaled positional encoding, dropout
    (speech_transformer.py:108-113)."""

    def __init__(self, store, num_class, d_model, rate):
        super().__init__()
        self.store, self.num_class, self.d_model, self.rate = store, num_class, d_model, rate
        store.register("y_net.embedding", (num_class, d_model), uniform(-0.05, 0.05))
        self.xscale = float(torch.sqrt(torch.tensor(float(d_model), dtype=torch.float32)))
        self.site_drop = store.new_site()
        self.embedding = None

    def bind(self):
        p = nn.Parameter(self.store.view("y_net.embedding"))
        p.grad = self.store.grad_view("y_net.embedding")
        self.embedding = p

    def forward(self, y0, training=None):
        return YNetFn.apply(y0, self, bool(training), self.embedding)


class SpeechTransformer(BaseModel):
    """x_net for the input, y_net for the shifted labels, the transformer and the output layer."""
    default_config = {
        "return_encoder_output": False,
        "num_filters": 512,
        "d_model": 512,
        "num_heads": 8,
        "num_encoder_layers": 12,
        "num_decoder_layers": 6,
        "dff": 1280,
        "rate": 0.1,
        "schedual_sampling_rate": 0.9,
        "label_smoothing_rate": 0.0,
        "unidirectional": False,
        "look_ahead": 0,
        "conv_module_kernel_size": 0
    }

    def __init__(self, data_descriptions, config=None, precision=None, store=None, seed=1, device=None):
        super().__init__()
        self.hparams = register_and_parse_hparams(self.default_config, config, cls=self.__class__)
        self.num_class = data_descriptions.num_class + 1
        self.sos = self.num_class - 1
        self.eos = self.num_class - 1
        ls_rate = self.hparams.label_smoothing_rate
        self.loss_function = Seq2SeqSparseCategoricalCrossentropy(
            num_classes=self.num_class, eos=self.eos, label_smoothing=ls_rate)
        self.metric = Seq2SeqSparseCategoricalAccuracy(eos=self.eos, name="Accuracy")
        self.data_descriptions = data_descriptions

        own_store = store is None
        if own_store:
            import os
            precision = precision or os.environ.get("ATHENA_PRECISION", "bf16")
            store = ParamStore(_PRECISIONS[precision], device=device)
        self.store = store
        hp = self.hparams
        self.x_net = XNet(store, data_descriptions.sample_shape["input"], hp.num_filters, hp.d_model, hp.rate)
        self.y_net = YNet(store, self.num_class, hp.d_model, hp.rate)
        self.transformer = Transformer(store, hp.d_model, hp.num_heads, hp.num_encoder_layers,
                                       hp.num_decoder_layers, hp.dff, hp.rate, unidirectional=hp.unidirectional,
                                       look_ahead=hp.look_ahead,
                                       conv_module_kernel_size=hp.conv_module_kernel_size)
        self.final_layer = Dense(store, "final_layer", hp.d_model, self.num_class)
        if own_store:
            self.finalize(seed)

    def finalize(self, seed=1):
        """Allocate the flat buffers and bind every Parameter view (once all variables are registered)."""
        self.store.finalize(seed)
        bind_all(self)
        self.y_net.bind()

    def prepare_samples(self, samples):
        return to_device_samples(samples, self.store.device)

    def call(self, samples, training=None):
        self.store.refresh_shadows()
        x0 = samples["input"]
        x = self.x_net(x0, training=training)
        # sos-prefixed labels, eos-terminated targets (kept for the loss), frame counts and both key
        # masks from one kernel; the reference's separate steps are the fallback for other batch forms
        prep = StepLabels.build(samples, x.shape[1], self.sos, self.eos)
        y0 = prep.y0 if prep is not None else insert_sos_in_labels(samples["output"], self.sos)
        y = self.y_net(y0, training=training)
        if self.store.collect_cuts:
            self.store.cuts = [("y_net", y)]
        if prep is not None:
            input_mask, output_mask = AttnMask(prep.x_mask, False), AttnMask(prep.y_mask, True)
        else:
            input_length = self.compute_logit_length(samples)
            input_mask, output_mask = create_multihead_mask(x, input_length, y0)
        y, encoder_output = self.transformer(x, y, input_mask, output_mask, input_mask, training=training,
                                             return_encoder_output=True)
        if self.store.collect_cuts:
            self.store.cuts.append(("memory", encoder_output))
        y = self._project(self.final_layer, y)
        if self.hparams.return_encoder_output:
            return y, encoder_output
        return y

    @staticmethod
    def _project(dense, h, joins_memory_grad=False):
        """Dense(V): fp32 logits [B, T, V] as a view of rows padded to a multiple of 8 columns.
        ``joins_memory_grad``: ``h`` is the encoder output the decoder layers attend to, and the
        caller's total loss reaches it through them as well."""
        y = DenseFn.apply(h, dense, dense.kernel, dense.bias, joins_memory_grad)
        # the projection's backward pass consumes bf16: the loss kernels may write their gradient so
        y._athena_accepts_bf16_grad = h.dtype == torch.bfloat16
        return y

    def compute_logit_length(self, samples):
        """ceil(ceil(len/2)/2) (speech_transformer.py:156-162)."""
        src = samples["input_length"]
        prepared = StepLabels.logit_length_for(src)
        if prepared is not None:
            return prepared
        hit = getattr(self, "_logit_length_cache", None)
        if hit is not None and hit[0] is src and hit[1] == src._version:
            return hit[2]        # asked for again by the loss with the same batch
        n = src.to(torch.int32)
        out = (((n + 1) // 2) + 1) // 2
        self._logit_length_cache = (src, src._version, out)
        return out

    def restore_from_pretrained_model(self, pretrained_model, model_type=""):
        if model_type == "":
            return
        if model_type in ("mpc", "SpeechTransformer"):
            logging.info("loading from pretrained %s model", model_type)
            state = pretrained_model.store.state_dict()
            prefixes = ("x_net.", "encoder.") if model_type == "mpc" else \
                ("x_net.", "y_net.", "encoder.", "decoder.", "final_layer.")
            sub = {k: v for k, v in state.items() if k.startswith(prefixes) and k in self.store.by_name}
            self.store.load_state_dict(sub, strict=False)
        else:
            raise ValueError("NOT SUPPORTED")

    def time_propagate(self, history_logits, history_predictions, step, enc_outputs):
        """One decoding step for every live hypothesis (speech_transformer.py:164-194): the decoder
        runs over the whole history ``history_predictions`` [n, step + 1] (int, <sos> first) and the
        logits of the newest position come back.  Returns (logits [n, num_class] fp32,
        history_logits + [logits], step + 1)."""
        encoder_output, memory_mask = enc_outputs
        if isinstance(history_predictions, (list, tuple)):
            history_predictions = torch.stack([torch.as_tensor(t) for t in history_predictions], dim=1)
        y0 = torch.as_tensor(history_predictions).to(device=self.store.device, dtype=torch.int32).contiguous()
        n = y0.shape[0]
        memory = encoder_output
        if memory.shape[0] != n:
            memory = memory[:1].expand(n, -1, -1)
        memory = memory.contiguous()
        mm = AttnMask.coerce(memory_mask)
        if mm.key_mask is not None and mm.key_mask.shape[0] != n:
            mm = AttnMask(mm.key_mask[:1].expand(n, -1).contiguous(), False)
        with torch.no_grad():
            self.store.refresh_shadows()
            y = self.y_net(y0, training=False)
            y = self.transformer.decoder(y, memory, tgt_mask=AttnMask(None, True), memory_mask=mm, training=False)
            logits = self._project(self.final_layer, y)[:, -1, :].to(torch.float32)
        return logits, list(history_logits) + [logits], step + 1

    def decode(self, samples, hparams, decoder, return_encoder=False):
        """Beam-search decoding of a batch of one utterance (speech_transformer.py:196-238): the
        encoder runs once in inference mode; with ``return_encoder`` its output and the input mask
        are returned, otherwise ``decoder`` searches from <sos> and its predictions come back."""
        with torch.no_grad():
            self.store.refresh_shadows()
            x = self.x_net(samples["input"], training=False)
            input_length = self.compute_logit_length(samples)
            input_mask, _ = create_multihead_mask(x, input_length, None)
            encoder_output = self.transformer.encoder(x, input_mask, training=False)
        if return_encoder:
            return encoder_output, input_mask
        batch = samples["input"].shape[0]
        history_predictions = torch.full((batch, 1), self.sos, dtype=torch.int64)
        return decoder(history_predictions, [history_predictions], 0, (encoder_output, input_mask))
