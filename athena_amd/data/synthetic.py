"""A dataset builder of synthetic utterances with the interface the training entry point uses.

It stands where the reference's ``SpeechRecognitionDatasetBuilder``
(athena/data/datasets/speech_recognition.py:25-206, athena/data/datasets/base.py:29-161) stands:
``num_class``, ``sample_shape`` / ``sample_type`` / ``sample_signature``, ``hparams`` (text_config,
words), ``shard``, ``batch_wise_shuffle``, ``compute_cmvn_if_necessary`` and ``as_dataset``, which
yields the same ``samples`` dicts (``input`` [B, T, F, channels] fp32 zero-padded, ``input_length``,
``output`` [B, U] int32 zero-padded, ``output_length``).  The utterances are seeded random
features and labels, sorted by length as the reference does for its first epochs ("sorta").
CSV / vocabulary / CMVN reading of real corpora is not part of this package.
"""
import numpy as np
import torch

from ..utils.hparam import register_and_parse_hparams


class SyntheticSpeechRecognitionDatasetBuilder:
    default_config = {
        "num_utterances": 64,
        "min_frames": 100,
        "max_frames": 200,
        "feat_dim": 40,
        "num_channels": 1,
        "vocab_size": 30,          # label ids 1 .. vocab_size - 1; 0 is padding / <unk>
        "min_labels": 3,
        "max_labels": 8,
        "seed": 1,
        "text_config": {"type": "vocab", "model": None},
        "words": None,
    }

    def __init__(self, config=None):
        self.hparams = register_and_parse_hparams(self.default_config, config, cls=self.__class__)
        p = self.hparams
        rng = np.random.default_rng(p.seed)
        self.entries = []
        for _ in range(p.num_utterances):
            frames = int(rng.integers(p.min_frames, p.max_frames + 1))
            # every utterance keeps room for its labels after the two stride-2 convolutions
            n_lab = int(rng.integers(p.min_labels, min(p.max_labels, max(frames // 8, p.min_labels)) + 1))
            feat = rng.standard_normal((frames, p.feat_dim, p.num_channels)).astype(np.float32)
            labels = rng.integers(1, p.vocab_size, size=n_lab).astype(np.int32)
            self.entries.append((feat, labels))
        self.entries.sort(key=lambda e: e[0].shape[0])              # "sorta": by duration
        self.order = list(range(len(self.entries)))

    # ---- what the models read ------------------------------------------------------------------
    @property
    def num_class(self):
        return self.hparams.vocab_size

    @property
    def sample_type(self):
        return {"input": torch.float32, "input_length": torch.int32, "output": torch.int32,
                "output_length": torch.int32}

    @property
    def sample_shape(self):
        p = self.hparams
        return {"input": (None, p.feat_dim, p.num_channels), "input_length": (), "output": (None,),
                "output_length": ()}

    @property
    def sample_signature(self):
        p = self.hparams
        return ({"input": ((None, None, p.feat_dim, p.num_channels), torch.float32),
                 "input_length": ((None,), torch.int32), "output": ((None, None), torch.int32),
                 "output_length": ((None,), torch.int32)},)

    # ---- what the training loop calls (base.py:117-161) ------------------------------------------
    def __len__(self):
        return len(self.order)

    def shard(self, num_shards, index):
        """Keep every ``num_shards``-th utterance starting at ``index``, after trimming the list to a
        multiple of ``num_shards`` so that every rank gets the same count and therefore the same
        number of steps (and collectives) per epoch (base.py:126-138)."""
        if num_shards <= 1:
            return self
        if not 0 <= index < num_shards:
            raise ValueError("shard: index %d outside [0, %d)" % (index, num_shards))
        total = (len(self.order) // num_shards) * num_shards
        self.order = self.order[:total][index::num_shards]
        return self

    def batch_wise_shuffle(self, batch_size=64):
        """Shuffle whole batches, keeping utterances of similar length together (base.py:140-161)."""
        if len(self.order) <= batch_size:
            return self
        rng = np.random.default_rng(self.hparams.seed + len(self.order))
        nb = len(self.order) // batch_size
        head, tail = self.order[:nb * batch_size], self.order[nb * batch_size:]
        batches = [head[i * batch_size:(i + 1) * batch_size] for i in range(nb)]
        rng.shuffle(batches)
        self.order = [i for b in batches for i in b] + tail
        return self

    def compute_cmvn_if_necessary(self, is_necessary=True):
        """The synthetic features are unit-variance already."""
        return self

    def as_dataset(self, batch_size=16, num_threads=1):
        """List of padded batches (``samples`` dicts of CPU tensors)."""
        out = []
        for i in range(0, len(self.order), batch_size):
            chunk = [self.entries[j] for j in self.order[i:i + batch_size]]
            T = max(f.shape[0] for f, _ in chunk)
            U = max(l.shape[0] for _, l in chunk)
            p = self.hparams
            x = np.zeros((len(chunk), T, p.feat_dim, p.num_channels), np.float32)
            y = np.zeros((len(chunk), U), np.int32)
            for k, (f, l) in enumerate(chunk):
                x[k, :f.shape[0]] = f
                y[k, :l.shape[0]] = l
            out.append({"input": torch.from_numpy(x),
                        "input_length": torch.tensor([f.shape[0] for f, _ in chunk], dtype=torch.int32),
                        "output": torch.from_numpy(y),
                        "output_length": torch.tensor([l.shape[0] for _, l in chunk], dtype=torch.int32)})
        return out


# ---- fixed-shape synthetic batches and model configurations (benchmark, smoke test, parity tests) ----
class DataDescriptions:
    """The part of a dataset builder that the models read (reference
    athena/data/datasets/speech_recognition.py:134-206): ``num_class``, ``sample_shape`` and
    ``hparams.text_config`` / ``hparams.words``."""

    class _HP:
        def __init__(self):
            self.text_config = {"type": "vocab", "model": None}
            self.words = None

    def __init__(self, num_class, feat_dim, in_channels=1):
        self.num_class = num_class
        self.sample_shape = {"input": (None, feat_dim, in_channels), "input_length": (), "output": (None,),
                             "output_length": ()}
        self.hparams = self._HP()


def model_config(d_model, num_filters, heads, enc, dec, dff, rate=0.0, mtl_weight=0.5, ls=0.0, **extra):
    """``model_config`` of ``mtl_transformer_ctc`` in the shape of the shipped recipes
    (examples/asr/*/configs/mtl_transformer_sp.json); ``extra`` goes into the inner model_config."""
    inner = {"return_encoder_output": True, "num_filters": num_filters, "d_model": d_model, "num_heads": heads,
             "num_encoder_layers": enc, "num_decoder_layers": dec, "dff": dff, "rate": rate,
             "label_smoothing_rate": ls}
    inner.update(extra)
    return {"model": "speech_transformer", "model_config": inner, "mtl_weight": mtl_weight}


def make_samples(B, T, feat_dim, V, U, seed=1, full_length=True, dtype=torch.float32, in_channels=1):
    """One synthetic batch of the ``samples`` dict: unit-variance features zeroed beyond
    ``input_length`` (all T frames, or lengths drawn from [0.6 T, T] and sorted longest first), label
    ids uniform in [1, V-2] zero-padded beyond ``output_length`` (drawn from [U/2, U]; the first
    utterance has U labels)."""
    rng = np.random.default_rng(seed)
    feats = rng.standard_normal((B, T, feat_dim, in_channels)).astype(np.float32)
    if full_length:
        frames = np.full((B,), T, dtype=np.int32)
    else:
        frames = np.sort(rng.integers(int(0.6 * T), T + 1, size=B).astype(np.int32))[::-1].copy()
        frames[0] = T
    for b in range(B):
        feats[b, frames[b]:] = 0.0
    n_labels = rng.integers(max(U // 2, 1), U + 1, size=B).astype(np.int32)
    n_labels[0] = U
    ids = rng.integers(1, max(V - 1, 2), size=(B, U)).astype(np.int32)
    for b in range(B):
        ids[b, n_labels[b]:] = 0
    return {"input": torch.from_numpy(feats).to(dtype), "input_length": torch.from_numpy(frames),
            "output": torch.from_numpy(ids), "output_length": torch.from_numpy(n_labels)}
