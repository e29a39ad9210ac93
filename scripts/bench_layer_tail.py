"""Time the forward pass of one encoder layer with the one-launch layer tail on and off."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd.layers.commons import bind_all
from athena_amd.layers.transformer import TransformerEncoderLayer
from athena_amd.runtime import ParamStore
from scripts.bench_kernels import timeit, dev

B, T, d, dff = 32, 250, 256, 1024
store = ParamStore(torch.bfloat16, device=dev)
layer = TransformerEncoderLayer(store, "enc", d, 4, dff, 0.1)
store.finalize(seed=1)
bind_all(layer)
store.refresh_shadows(force=True)
x = torch.randn(B, T, d, device=dev).to(torch.bfloat16).requires_grad_(True)
dy = torch.randn(B, T, d, device=dev).to(torch.bfloat16)
for flag in ("0", "1"):
    os.environ["ATHENA_LAYER_TAIL"] = flag
    def fwd():
        with torch.no_grad():
            layer(x, None, training=True)
    tf = timeit(fwd, reps=10)
    print("layer tail %s: forward %.1f us" % (flag, tf * 1e6))
