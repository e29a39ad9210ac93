"""Where the fill launches of a training step come from: one eager step with torch's zero-filling
calls wrapped so that each prints the call site."""
import collections
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

wl = dict(bench.WORKLOADS["aishell_b32_t1000"], precision="bf16")
model, opt, solver, mc = bench.build(wl, "bf16", seed=1)
solver.hparams.enable_tf_function = False
from athena_amd.data import make_samples  # noqa: E402
ds = model.prepare_samples(make_samples(wl["B"], wl["T"], wl["feat"], wl["vocab"] + 1, wl["U"], seed=1, full_length=True))
solver.train_step(ds)
torch.cuda.synchronize()
sites = collections.Counter()


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        st = [fr for fr in traceback.extract_stack()[:-1] if "athena_amd" in fr.filename or "bench.py" in fr.filename]
        if st:
            fr = st[-1]
            sites["%s @ %s:%d" % (name, os.path.relpath(fr.filename), fr.lineno)] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


for n in ("zeros", "zeros_like", "full", "ones"):
    wrap(torch, n)
for n in ("zero_", "fill_", "new_zeros", "clone", "contiguous", "copy_"):
    wrap(torch.Tensor, n)
solver.train_step(ds)
torch.cuda.synchronize()
for k, v in sorted(sites.items()):
    print(v, k)
