"""Where the bf16 mode's logit error comes from: the encoder output of the bf16 mode against the fp32
mode, and the CTC head's logits computed (a) by the bf16 mode, (b) in fp32 from the bf16 mode's encoder
output with the fp32 weights -- what a projection without any rounding of its own would give."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from athena_amd import MtlTransformerCtc  # noqa: E402
from athena_amd.data import DataDescriptions, make_samples, model_config  # noqa: E402

wl = bench.WORKLOADS["aishell_b32_t1000"]
mc = model_config(wl["d_model"], wl["num_filters"], wl["heads"], wl["enc"], wl["dec"], wl["dff"], 0.0,
                  mtl_weight=wl.get("mtl_weight", 0.5))
samples = make_samples(wl["B"], wl["T"], wl["feat"], wl["vocab"] + 1, wl["U"], seed=1, full_length=True)
res = {}
for prec in ("fp32", "bf16"):
    m = MtlTransformerCtc(DataDescriptions(wl["vocab"], wl["feat"]), copy.deepcopy(mc), precision=prec, seed=1)
    with torch.no_grad():
        s = m.prepare_samples(samples)
        att, mem = m.model(s, training=True)
        from athena_amd.models.speech_transformer import SpeechTransformer
        ctc = SpeechTransformer._project(m.decoder, mem)
        res[prec] = dict(att=att.float().cpu(), mem=mem.float().cpu(), ctc=ctc.float().cpu(),
                         W=m.decoder.kernel.detach().float().cpu(), b=m.decoder.bias.detach().float().cpu())
    del m
f, h = res["fp32"], res["bf16"]
V = f["W"].shape[1]


def err(a, b, what):
    d = (a - b).double()
    print("%-58s rms %.3e  max %.3e  (scale %.2f)" % (what, float(d.pow(2).mean().sqrt()), float(d.abs().max()),
                                                     float(b.abs().max())))


err(h["mem"], f["mem"], "encoder output, bf16 mode vs fp32 mode")
err(h["ctc"][..., :V], f["ctc"][..., :V], "CTC logits, bf16 mode vs fp32 mode")
c_exact = h["mem"].reshape(-1, h["mem"].shape[-1]) @ f["W"] + f["b"]
err(c_exact, f["ctc"][..., :V].reshape(-1, V), "CTC logits from the bf16 mode's encoder output, exact head")
w_b = f["W"].bfloat16().float()
c_wb = h["mem"].reshape(-1, h["mem"].shape[-1]) @ w_b + f["b"]
err(c_wb, f["ctc"][..., :V].reshape(-1, V), "the same with the head's weights rounded to bf16")
err(h["att"], f["att"], "attention logits, bf16 mode vs fp32 mode")
# what the rounding of the head's input alone costs: the fp32 mode's encoder output rounded to bf16, exact head
c_hr = f["mem"].bfloat16().float().reshape(-1, f["mem"].shape[-1]) @ f["W"] + f["b"]
err(c_hr, f["ctc"][..., :V].reshape(-1, V), "exact head on the fp32 mode's encoder output rounded to bf16")
c_wr = f["mem"].reshape(-1, f["mem"].shape[-1]) @ w_b + f["b"]
err(c_wr, f["ctc"][..., :V].reshape(-1, V), "fp32 encoder output, head's weights rounded to bf16")
