"""Speech-recognition data set from a CSV index (mirror of the reference's
athena/data/datasets/speech_recognition.py:25-206 and athena/data/datasets/base.py:29-211).

``data_csv``: tab-separated, header ``wav_filename  wav_length_ms  transcript [speaker]``.  The
builder keeps the reference's order of operations: speakers, speed permutation, sort by duration,
vocabulary fitted on the transcripts when the text model is "text", then the unknown-token and
length filters.  ``as_dataset`` yields the padded ``samples`` dicts the models and solvers consume
(incomplete last batch dropped, as with ``padded_batch(drop_remainder=True)``).
"""
import logging
import math
import os
import random

import numpy as np
import torch

from ..utils.hparam import register_and_parse_hparams
from .audio_featurizer import AudioFeaturizer
from .feature_normalizer import FeatureNormalizer
from .text_featurizer import TextFeaturizer


class BaseDatasetBuilder:
    """entries + hparams + the sharding / shuffling every data set shares (base.py:84-161)."""
    default_config = {}

    def __init__(self, config=None):
        self.hparams = register_and_parse_hparams(self.default_config, config, cls=self.__class__)
        logging.info("hparams: %s", self.hparams)
        self.entries = []

    def reload_config(self, config):
        if config is not None:
            self.hparams.override_from_dict(config)

    def preprocess_data(self, file_path):
        raise NotImplementedError

    def __getitem__(self, index):
        raise NotImplementedError

    def __len__(self):
        return len(self.entries)

    def shard(self, num_shards, index):
        """Keep every ``num_shards``-th entry from ``index`` of the first
        (len // num_shards) * num_shards entries: every rank gets the same count."""
        if index >= num_shards or index < 0:
            raise ValueError("the index should smaller the num_shards")
        logging.info("Creates the sub-dataset which is the %d part of %d", index, num_shards)
        total = (len(self.entries) // num_shards) * num_shards
        self.entries = self.entries[:total][index::num_shards]
        return self

    def batch_wise_shuffle(self, batch_size=64):
        """Shuffle whole batches of the (duration-sorted) entries; the incomplete tail stays last."""
        if len(self.entries) == 0:
            return self
        logging.info("perform batch_wise_shuffle with batch_size %d", batch_size)
        buckets = list(range(int(math.floor(len(self.entries) / batch_size))))
        random.shuffle(buckets)
        shuffled = []
        for i in buckets:
            shuffled.extend(self.entries[i * batch_size:(i + 1) * batch_size])
        shuffled.extend(self.entries[len(buckets) * batch_size:])
        self.entries = shuffled
        return self

    def compute_cmvn_if_necessary(self, is_necessary=True):
        return self


class SpeechRecognitionDatasetBuilder(BaseDatasetBuilder):
    default_config = {
        "audio_config": {"type": "Fbank"},
        "text_config": {"type": "vocab", "model": "athena/utils/vocabs/ch-en.vocab"},
        "num_cmvn_workers": 1,
        "cmvn_file": None,
        "remove_unk": True,
        "input_length_range": [20, 50000],
        "output_length_range": [1, 10000],
        "speed_permutation": [1.0],
        "data_csv": None,
        "words": None
    }

    def __init__(self, config=None):
        super().__init__(config=config)
        self.speakers = []
        self.audio_featurizer = AudioFeaturizer(self.hparams.audio_config)
        self.feature_normalizer = FeatureNormalizer(self.hparams.cmvn_file)
        self.text_featurizer = TextFeaturizer(self.hparams.text_config)
        if self.hparams.data_csv is not None:
            self.preprocess_data(self.hparams.data_csv)

    def preprocess_data(self, file_path):
        """entries = (audio, duration / speed, transcript, speed, speaker), sorted by duration and
        filtered (speech_recognition.py:46-102)."""
        logging.info("Loading data from %s", file_path)
        with open(file_path, "r", encoding="utf-8") as f:
            lines = f.read().splitlines()
        headers = lines[0].split("\t")
        rows = [tuple(line.split("\t")) for line in lines[1:]]
        self.speakers = []
        if "speaker" not in headers:
            rows = [(wav, length, text, "global") for wav, length, text in rows]
            self.speakers.append("global")
        else:
            for _, _, _, speaker in rows:
                if speaker not in self.speakers:
                    self.speakers.append(speaker)
        if len(self.hparams.speed_permutation) > 1:
            logging.info("perform speed permutation")
        entries = []
        for speed in self.hparams.speed_permutation:
            for wav, length, text, speaker in rows:
                entries.append((wav, float(length) / float(speed), text, speed, speaker))
        entries.sort(key=lambda item: float(item[1]))
        if self.text_featurizer.model_type == "text":
            self.text_featurizer.load_model([e[2] for e in entries])
        unk = self.text_featurizer.unk_index
        if self.hparams.remove_unk and unk != -1:
            entries = [e for e in entries if unk not in self.text_featurizer.encode(e[2])]
        lo, hi = self.hparams.input_length_range
        entries = [e for e in entries if lo <= int(e[1]) < hi]
        lo, hi = self.hparams.output_length_range
        entries = [e for e in entries if lo <= len(e[2]) < hi]
        self.entries = entries
        return self

    def __getitem__(self, index):
        audio, _, transcript, speed, speaker = self.entries[index]
        feat = self.audio_featurizer(audio, speed=speed)
        feat = self.feature_normalizer(feat, speaker)
        label = self.text_featurizer.encode(transcript)
        return {"input": feat, "input_length": feat.shape[0], "output_length": len(label), "output": label}

    @property
    def num_class(self):
        """the max index of the vocabulary + 1"""
        return len(self.text_featurizer)

    @property
    def sample_type(self):
        return {"input": torch.float32, "input_length": torch.int32, "output_length": torch.int32,
                "output": torch.int32}

    @property
    def sample_shape(self):
        dim, nc = self.audio_featurizer.dim, self.audio_featurizer.num_channels
        return {"input": (None, dim, nc), "input_length": (), "output_length": (), "output": (None,)}

    @property
    def sample_signature(self):
        dim, nc = self.audio_featurizer.dim, self.audio_featurizer.num_channels
        return ({"input": ((None, None, dim, nc), torch.float32), "input_length": ((None,), torch.int32),
                 "output_length": ((None,), torch.int32), "output": ((None, None), torch.int32)},)

    def compute_cmvn_if_necessary(self, is_necessary=True):
        """Compute and save the per-speaker statistics unless the file exists (base.py:179-194)."""
        if not is_necessary:
            return self
        if self.hparams.cmvn_file is None:
            raise ValueError("compute_cmvn_if_necessary: cmvn_file is not set")
        if os.path.exists(self.hparams.cmvn_file):
            return self
        feature_dim = self.audio_featurizer.dim * self.audio_featurizer.num_channels
        self.feature_normalizer.compute_cmvn(self.entries, self.speakers, self.audio_featurizer, feature_dim,
                                             self.hparams.num_cmvn_workers)
        self.feature_normalizer.save_cmvn(["speaker", "mean", "var"])
        return self

    def as_dataset(self, batch_size=16, num_threads=1):
        """Batches of ``batch_size`` consecutive entries, features and labels zero-padded to the
        longest of the batch (base.py:29-81); the incomplete last batch is dropped.  The result can
        be iterated any number of times; every pass computes its batches as it goes, ``num_threads``
        workers ahead of the consumer (the reference: a generator with padded_batch and prefetch)."""
        if len(self) == 0:
            raise ValueError("num samples is empty")
        return _BatchStream(self, batch_size, num_threads)

    def make_batch(self, first, batch_size):
        """Entries first .. first + batch_size - 1 as one zero-padded batch."""
        dim, nc = self.audio_featurizer.dim, self.audio_featurizer.num_channels
        items = [self[i] for i in range(first, first + batch_size)]
        T = max(it["input_length"] for it in items)
        U = max(it["output_length"] for it in items)
        x = np.zeros((batch_size, T, dim, nc), dtype=np.float32)
        y = np.zeros((batch_size, U), dtype=np.int32)
        for k, it in enumerate(items):
            x[k, :it["input_length"]] = it["input"]
            y[k, :it["output_length"]] = it["output"]
        return {"input": torch.from_numpy(x),
                "input_length": torch.tensor([it["input_length"] for it in items], dtype=torch.int32),
                "output": torch.from_numpy(y),
                "output_length": torch.tensor([it["output_length"] for it in items], dtype=torch.int32)}


class _BatchStream:
    """The batches of one epoch, made on demand.  A pass over it holds at most ``depth`` finished
    batches: worker threads (feature extraction spends its time in numpy, which releases the
    interpreter lock) fill a bounded window in batch order, the consumer takes them in that order.
    The entry list is read when a pass starts, so a ``batch_wise_shuffle`` between epochs is seen."""

    def __init__(self, builder, batch_size, num_threads=1, depth=None):
        self.builder, self.batch_size = builder, int(batch_size)
        self.workers = max(int(num_threads or 1), 1)
        self.depth = int(depth) if depth else 2 * self.workers

    def __len__(self):
        return len(self.builder) // self.batch_size

    def __iter__(self):
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor
        n, bs, make = len(self), self.batch_size, self.builder.make_batch
        if self.workers == 1 and self.depth <= 1:
            for b in range(n):
                yield make(b * bs, bs)
            return
        pool = ThreadPoolExecutor(max_workers=self.workers)
        window, nxt = deque(), 0
        try:
            while nxt < n or window:
                while nxt < n and len(window) < self.depth:
                    window.append(pool.submit(make, nxt * bs, bs))
                    nxt += 1
                yield window.popleft().result()
        finally:
            for f in window:
                f.cancel()
            pool.shutdown(wait=True)
