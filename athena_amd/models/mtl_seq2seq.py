"""Multi-task attention + CTC model (mirror of the reference's athena/models/mtl_seq2seq.py:32-115)."""
import io
import os

import torch

from ..layers.commons import Dense, bind_all
from ..loss import CTCLoss, weighted_total
from ..metrics import CTCAccuracy
from ..runtime import ParamStore
from ..utils.hparam import register_and_parse_hparams
from .base import BaseModel, LazyMetric, to_device_samples
from .speech_transformer import _PRECISIONS, SpeechTransformer


class MtlTransformerCtc(BaseModel):
    """Attention-based seq2seq model with an auxiliary CTC head on the encoder output:
    loss = mtl_weight * CE + (1 - mtl_weight) * CTC."""

    SUPPORTED_MODEL = {
        "speech_transformer": SpeechTransformer,
    }

    default_config = {
        "model": "speech_transformer",
        "model_config": {"return_encoder_output": True},
        "mtl_weight": 0.5
    }

    def __init__(self, data_descriptions, config=None, precision=None, seed=1, device=None):
        super().__init__()
        self.num_class = data_descriptions.num_class + 1
        self.sos = self.num_class - 1
        self.eos = self.num_class - 1
        self.hparams = register_and_parse_hparams(self.default_config, config, cls=self.__class__)
        self.loss_function = CTCLoss(blank_index=-1)
        self.metric = CTCAccuracy()
        if self.hparams.model not in self.SUPPORTED_MODEL:
            raise NotImplementedError("model %r is not built (speech_transformer2 is a decode-time variant)" %
                                      self.hparams.model)
        precision = precision or os.environ.get("ATHENA_PRECISION", "bf16")
        self.store = ParamStore(_PRECISIONS[precision], device=device)
        self.model = self.SUPPORTED_MODEL[self.hparams.model](
            data_descriptions, self.hparams.model_config, store=self.store)
        self.time_propagate = self.model.time_propagate
        self.decoder = Dense(self.store, "ctc_decoder", self.model.hparams.d_model, self.num_class)
        self.store.finalize(seed)
        bind_all(self)
        self.model.y_net.bind()

        # vocabulary for WFST decoding (mtl_seq2seq.py:68-78); binary (sentencepiece) models are skipped
        self.vocab = {}
        hp = getattr(data_descriptions, "hparams", None)
        text_config = getattr(hp, "text_config", None) if hp is not None else None
        path = text_config.get("model") if isinstance(text_config, dict) else None
        if path and os.path.exists(path):
            try:
                for line in io.open(path, "r", encoding="utf-8").readlines():
                    parts = line.strip().split()
                    if len(parts) >= 2:
                        self.vocab[parts[0]] = int(parts[1])
            except (UnicodeDecodeError, ValueError):
                self.vocab = {}

    def prepare_samples(self, samples):
        return to_device_samples(samples, self.store.device)

    def call(self, samples, training=None):
        attention_logits, encoder_output = self.model(samples, training=training)
        # the encoder output also feeds the attention decoder: its gradient from here joins theirs
        ctc_logits = SpeechTransformer._project(self.decoder, encoder_output, joins_memory_grad=True)
        return attention_logits, ctc_logits

    def get_loss(self, outputs, samples, training=None):
        attention_logits, ctc_logits = outputs
        logit_length = self.compute_logit_length(samples)
        # the CTC term enters the total with weight 1 - mtl_weight: that is the factor its gradient
        # will arrive with, so the loss can leave the gradient in the same pass over the logits
        ctc_weight = 1.0 - float(self.hparams.mtl_weight)
        extra_loss = self.loss_function(ctc_logits, samples, logit_length, upstream=ctc_weight if ctc_weight > 0 else None,
                                        exact=True)
        self.metric(ctc_logits, samples, logit_length)
        main_loss, metrics = self.model.get_loss(attention_logits, samples, training=training)
        mtl_weight = self.hparams.mtl_weight
        # mtl_weight * main + (1 - mtl_weight) * extra as one kernel (mtl_seq2seq.py:95)
        loss = weighted_total([main_loss, extra_loss], [mtl_weight, 1.0 - mtl_weight])
        metrics[self.metric.name] = LazyMetric(self.metric)
        return loss, metrics

    def compute_logit_length(self, samples):
        return self.model.compute_logit_length(samples)

    def reset_metrics(self):
        self.metric.reset_states()
        self.model.reset_metrics()

    def restore_from_pretrained_model(self, pretrained_model, model_type=""):
        self.model.restore_from_pretrained_model(pretrained_model, model_type)

    def decode(self, samples, hparams, decoder):
        """Joint attention / CTC decoding of one utterance (mtl_seq2seq.py:117-156): the encoder output
        feeds the attention decoder inside ``decoder`` and -- with a CTC weight -- the CTC head, whose
        log-probabilities initialise the prefix scorer."""
        encoder_output, input_mask = self.model.decode(samples, hparams, decoder, return_encoder=True)
        history_predictions = torch.full((1, 1), self.sos, dtype=torch.int64)
        init_cand_states = [history_predictions]
        if hparams.decoder_type == "beam_search_decoder":
            if hparams.ctc_weight != 0 and decoder.ctc_scorer is not None:
                with torch.no_grad():
                    ctc_logits = SpeechTransformer._project(self.decoder, encoder_output).to(torch.float32)
                    ctc_logits = torch.log_softmax(ctc_logits, dim=-1)
                init_cand_states = decoder.ctc_scorer.initial_state(init_cand_states, ctc_logits)
            return decoder(history_predictions, init_cand_states, 0, (encoder_output, input_mask))
        raise NotImplementedError("decoder_type %r: only beam_search_decoder is built (the WFST decoder is not)"
                                  % (hparams.decoder_type,))
