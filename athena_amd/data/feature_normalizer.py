"""Per-speaker cepstral mean / variance normalisation (mirror of the reference's
athena/data/feature_normalizer.py:31-213): statistics in a tab-separated file with one row per
speaker (``speaker``, ``mean``, ``var`` as JSON lists), applied as (x - mean) / sqrt(var)."""
import csv
import json
import logging
import os
import time

import numpy as np


class FeatureNormalizer:
    def __init__(self, cmvn_file=None):
        self.cmvn_file = cmvn_file
        self.cmvn_dict = {}
        self.speakers = []
        if cmvn_file is not None:
            self.load_cmvn()

    def __call__(self, feat_data, speaker, reverse=False):
        return self.apply_cmvn(feat_data, speaker, reverse=reverse)

    def apply_cmvn(self, feat_data, speaker, reverse=False):
        """Normalise features [T, dim, channels] (or back with ``reverse``); a speaker without
        statistics passes through unchanged."""
        if speaker not in self.cmvn_dict:
            return feat_data
        feat = np.asarray(feat_data, dtype=np.float32)
        shape = feat.shape[1:]
        mean = np.asarray(self.cmvn_dict[speaker][0], dtype=np.float32).reshape(shape)
        var = np.asarray(self.cmvn_dict[speaker][1], dtype=np.float32).reshape(shape)
        if reverse:
            return feat * np.sqrt(var) + mean
        return (feat - mean) / np.sqrt(var)

    def compute_cmvn(self, entries, speakers, featurizer, feature_dim, num_cmvn_workers=1):
        """Mean and variance of every speaker's frames from running sums (float32, as the
        reference accumulates them).  ``entries``: tuples whose first item is the audio and whose
        last item is the speaker.  The worker count is accepted for the interface; the sums are
        taken in this process."""
        start = time.time()
        for tar_speaker in speakers:
            total = np.zeros(feature_dim, dtype=np.float32)
            total2 = np.zeros(feature_dim, dtype=np.float32)
            frames = 0
            for items in entries:
                if items[-1] != tar_speaker:
                    continue
                feat = np.asarray(featurizer(items[0]), dtype=np.float32).reshape(-1, feature_dim)
                frames += feat.shape[0]
                total += feat.sum(axis=0, dtype=np.float32)
                total2 += np.square(feat).sum(axis=0, dtype=np.float32)
            if frames == 0:
                continue
            mean = total / np.float32(frames)
            var = total2 / np.float32(frames) - np.square(mean)
            self.cmvn_dict[tar_speaker] = (mean.tolist(), var.tolist())
        logging.info("finished compute cmvn, which cost %.4f s", time.time() - start)

    def load_cmvn(self):
        if not os.path.exists(self.cmvn_file):
            return
        with open(self.cmvn_file, "r", encoding="utf-8", newline="") as f:
            for row in csv.DictReader(f, delimiter="\t"):
                self.cmvn_dict[row["speaker"]] = (json.loads(row["mean"]), json.loads(row["var"]))
        logging.info("Successfully load cmvn file %s", self.cmvn_file)

    def save_cmvn(self, variable_list=("speaker", "mean", "var")):
        if self.cmvn_file is None:
            self.cmvn_file = os.path.expanduser("~/.athena/cmvn_file")
        cmvn_dir = os.path.dirname(self.cmvn_file)
        if cmvn_dir and not os.path.exists(cmvn_dir):
            os.makedirs(cmvn_dir)
        with open(self.cmvn_file, "w", encoding="utf-8", newline="") as f:
            w = csv.writer(f, delimiter="\t", lineterminator="\n")
            w.writerow(list(variable_list))
            for speaker, stats in self.cmvn_dict.items():
                w.writerow([speaker] + [json.dumps([float(v) for v in s]) for s in stats])
        logging.info("Successfully save cmvn file %s", self.cmvn_file)
