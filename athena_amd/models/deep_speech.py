"""Deep Speech 2 style CTC model (mirror of the reference's athena/models/deep_speech.py:29-96): two
wide-kernel convolutions with BatchNormalization + relu6, a stack of recurrent layers each followed by
BatchNormalization, Dense + relu6 and the output projection, trained with the CTC loss."""
import torch

from ..layers.commons import Dense, bind_all
from ..layers.recurrent import ACT_RELU6, ConvFront, DenseActFn, RnnBn
from ..loss import CTCLoss
from ..metrics import CTCAccuracy
from ..runtime import ParamStore
from ..utils.hparam import register_and_parse_hparams
from .base import BaseModel, to_device_samples
from .speech_transformer import _PRECISIONS, SpeechTransformer


class DeepSpeechModel(BaseModel):
    """a sample implementation of CTC model"""
    default_config = {
        "conv_filters": 256,
        "rnn_hidden_size": 1024,
        "num_rnn_layers": 6,
        "rnn_type": "gru"
    }

    def __init__(self, data_descriptions, config=None, precision=None, store=None, seed=1, device=None):
        super().__init__()
        self.num_classes = data_descriptions.num_class + 1
        self.loss_function = CTCLoss(blank_index=-1)
        self.metric = CTCAccuracy()
        self.hparams = register_and_parse_hparams(self.default_config, config, cls=self.__class__)
        own_store = store is None
        if own_store:
            import os
            precision = precision or os.environ.get("ATHENA_PRECISION", "bf16")
            store = ParamStore(_PRECISIONS[precision], device=device)
        self.store = store
        hp = self.hparams
        # variable names follow the order Keras creates the layers in (conv2d, batch_normalization,
        # conv2d_1, batch_normalization_1, rnn, batch_normalization_2, ..., dense, dense_1)
        self.front = ConvFront(store, data_descriptions.sample_shape["input"], hp.conv_filters)
        layers, width = [], self.front.out_dim
        for i in range(hp.num_rnn_layers):
            layers.append(RnnBn(store, i, hp.rnn_type, width, hp.rnn_hidden_size))
            width = hp.rnn_hidden_size
        self.rnn_layers = torch.nn.ModuleList(layers)
        self.dense = Dense(store, "dense", width, hp.rnn_hidden_size)
        self.final_layer = Dense(store, "dense_1", hp.rnn_hidden_size, self.num_classes)
        if own_store:
            self.store.finalize(seed)
            bind_all(self)

    def prepare_samples(self, samples):
        return to_device_samples(samples, self.store.device)

    def call(self, samples, training=None):
        self.store.refresh_shadows()
        x = self.front(samples["input"], training=training)
        for layer in self.rnn_layers:
            x = layer(x, training=training)
        x = DenseActFn.apply(x, self.dense, ACT_RELU6, self.dense.kernel, self.dense.bias)
        return SpeechTransformer._project(self.final_layer, x)

    def compute_logit_length(self, samples):
        """ceil(ceil(len / 2) / 2) (deep_speech.py:90-96)."""
        n = samples["input_length"].to(torch.int32)
        return (((n + 1) // 2) + 1) // 2
