"""Generates tests/golden/ctc_kat.json: CTC losses obtained by enumerating every alignment
(oracle.ctc.ctc_brute_force).  Run from the repository root: python tests/golden/make_ctc_kat.py"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.ctc import ctc_brute_force  # noqa: E402

rng = np.random.default_rng(2024)
cases = []
for (B, T, V, U) in [(2, 5, 4, 2), (3, 6, 4, 3), (2, 4, 3, 1), (2, 6, 5, 3)]:
    logits = np.round(rng.normal(size=(B, T, V)) * 2, 4)
    labels = rng.integers(0, V - 1, size=(B, U)).astype(int)   # class 0 may appear as a real label
    llen = rng.integers(1, U + 1, size=B)
    tlen = rng.integers(U + 2, T + 1, size=B)
    labels[0, :] = labels[0, 0]                                  # a sample made of one repeated label
    llen[0] = U
    tlen[0] = T
    losses = [ctc_brute_force(logits[b, :tlen[b]], labels[b, :llen[b]], V - 1) for b in range(B)]
    lab = labels.copy()
    for b in range(B):
        lab[b, llen[b]:] = 0
    if not all(np.isfinite(losses)):
        continue
    cases.append(dict(logits=logits.tolist(), labels=lab.tolist(), label_length=llen.tolist(),
                      logit_length=tlen.tolist(), loss=losses))
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ctc_kat.json")
json.dump(dict(generator="tests/golden/make_ctc_kat.py (oracle.ctc.ctc_brute_force: all alignments enumerated)",
               cases=cases), open(out, "w"))
print("wrote", out, len(cases), "cases")
