"""Small model configurations shared by the model-level tests, the smoke test and bench.py."""
import torch


class DataDescriptions:
    """The part of a dataset builder the models read (reference
    athena/data/datasets/speech_recognition.py:134-206): num_class, sample_shape, hparams."""

    class _HP:
        def __init__(self):
            self.text_config = {"type": "vocab", "model": None}
            self.words = None

    def __init__(self, num_class, feat_dim, in_channels=1):
        self.num_class = num_class
        self.sample_shape = {"input": (None, feat_dim, in_channels), "input_length": (), "output": (None,),
                             "output_length": ()}
        self.hparams = self._HP()


def model_config(d_model, num_filters, heads, enc, dec, dff, rate=0.0, mtl_weight=0.5, ls=0.0):
    return {"model": "speech_transformer",
            "model_config": {"return_encoder_output": True, "num_filters": num_filters, "d_model": d_model,
                             "num_heads": heads, "num_encoder_layers": enc, "num_decoder_layers": dec, "dff": dff,
                             "rate": rate, "label_smoothing_rate": ls},
            "mtl_weight": mtl_weight}


def oracle_cfg(mc, num_class_plus1, feat_dim):
    c = dict(mc["model_config"])
    c.update(num_class=num_class_plus1, feat_dim=feat_dim, mtl_weight=mc["mtl_weight"])
    return c


TINY = dict(vocab=10, feat=12, mc=model_config(32, 16, 4, 2, 1, 64), B=3, T=37, U=4)
SMALL = dict(vocab=30, feat=20, mc=model_config(64, 32, 2, 2, 2, 128), B=4, T=61, U=6)
