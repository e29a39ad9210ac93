"""Training from a CSV-indexed data set on the GPU through the entry point of the package (reference
athena/main.py:58-185 with dataset_builder "speech_recognition_dataset")."""
import copy
import json
import os

import numpy as np
import pytest

from tests.modelutil import TINY

pytestmark = pytest.mark.gpu
EN_VOCAB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vocabs", "en.vocab")


def _corpus(tmp_path, name, n, seed):
    rng = np.random.default_rng(seed)
    words = ["ab", "cab", "dead", "bee", "face", "a bad cab"]
    lines = ["wav_filename\twav_length_ms\ttranscript\tspeaker"]
    for i in range(n):
        frames = int(rng.integers(48, 90))
        path = str(tmp_path / ("%s%d.npy" % (name, i)))
        np.save(path, rng.standard_normal((frames, 12)).astype(np.float32))
        lines.append("%s\t%d\t%s\tspk%d" % (path, frames * 10, words[i % len(words)], i % 2))
    csv = str(tmp_path / (name + ".csv"))
    with open(csv, "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")
    return csv


def test_training_entry_point_on_a_csv_data_set(tmp_path):
    from athena_amd import BaseSolver
    from athena_amd import main as amain
    common = {"audio_config": {"type": "Fbank", "filterbank_channel_count": 12},
              "text_config": {"type": "vocab", "model": EN_VOCAB}, "cmvn_file": str(tmp_path / "cmvn")}
    cfg = {"batch_size": 4, "num_epochs": 2, "sorta_epoch": 1, "ckpt": str(tmp_path / "ckpts"),
           "summary_dir": str(tmp_path / "event"),
           "solver_config": {"clip_norm": 100.0, "log_interval": 1, "enable_tf_function": False},
           "model": "mtl_transformer_ctc", "model_config": copy.deepcopy(TINY["mc"]),
           "optimizer": "warmup_adam", "optimizer_config": {"d_model": 32, "warmup_steps": 5, "k": 0.5},
           "dataset_builder": "speech_recognition_dataset",
           "trainset_config": dict(common, data_csv=_corpus(tmp_path, "train", 14, 1)),
           "devset_config": dict(common, data_csv=_corpus(tmp_path, "dev", 6, 2))}
    path = tmp_path / "cfg.json"
    path.write_text(json.dumps(cfg))
    loss, metrics = amain.train(str(path), BaseSolver)
    assert loss == loss and set(metrics) == {"Accuracy", "CTCAccuracy"}
    assert os.path.exists(tmp_path / "cmvn")                   # computed from the training set on first use
    assert sorted(os.listdir(tmp_path / "ckpts"))[:2] == ["best_loss", "checkpoint"]
