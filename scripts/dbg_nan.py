"""Which gradient tensors of a full-size step hold non-finite values (GPU scratch driver)."""
import copy
import sys

import torch

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import MtlTransformerCtc
from athena_amd.data import DataDescriptions, make_samples, model_config

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
case = dict(vocab=8000, feat=80, mc=model_config(256, 256, 4, 12, 0, 1024, mtl_weight=0.0), B=B, T=1000, U=200)
model = MtlTransformerCtc(DataDescriptions(case["vocab"], case["feat"]), copy.deepcopy(case["mc"]), precision=prec, seed=1)
s = make_samples(case["B"], case["T"], case["feat"], case["vocab"] + 1, case["U"], seed=5, full_length=False)
ds = model.prepare_samples(s)
att, ctc = model(ds, training=True)
att.retain_grad(); ctc.retain_grad()
loss, _ = model.get_loss((att, ctc), ds, training=True)
print("loss", float(loss), "att finite", bool(torch.isfinite(att).all()), "ctc finite", bool(torch.isfinite(ctc).all()))
loss.backward()
print("dctc finite", bool(torch.isfinite(ctc.grad).all()), "datt finite", bool(torch.isfinite(att.grad).all()))
for slot in model.store.train_slots:
    g = model.store.grad_view(slot.name)
    if not bool(torch.isfinite(g).all()):
        print("non-finite:", slot.name, int((~torch.isfinite(g)).sum()), "of", g.numel())
print("done")
