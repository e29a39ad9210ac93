"""Kernel durations of the one-launch layer tails (and the attention kernels beside them) in a stack of
encoder layers at the AISHELL shape, from the library's timing sessions (kernel begin / end timestamps)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import _lib
_lib.LIB_PATH = os.environ.get("ATHENA_STAMPS_LIB", _lib.LIB_PATH)
from athena_amd.layers.commons import bind_all
from athena_amd.layers.transformer import TransformerEncoder, TransformerEncoderLayer
from athena_amd.runtime import ParamStore
from athena_amd import ops
from bench import KernelTimer

dev = torch.device("cuda", 0)
B, T, d, dff, L = 32, 250, 256, int(os.environ.get("DFF", "1024")), int(os.environ.get("LAYERS", "4"))
store = ParamStore(torch.bfloat16, device=dev)
layers = [TransformerEncoderLayer(store, "encoder.%d" % i, d, 4, dff, float(os.environ.get("RATE", "0.1"))) for i in range(L)]
store.finalize(seed=1)
enc = TransformerEncoder(layers)
bind_all(enc)
store.refresh_shadows(force=True)
x = torch.randn(B, T, d, device=dev).to(torch.bfloat16).requires_grad_(True)
dy = torch.randn(B, T, d, device=dev).to(torch.bfloat16)


def step():
    x.grad = None
    y = enc(x, None, training=True)
    y.backward(dy)


names = ["athena_layer_tail_fwd", "athena_layer_tail_bwd", "athena_mha_fwd", "athena_mha_bwd", "athena_gemm",
         "athena_gemm_grouped", "athena_gemm_wgrad_table"]
for _ in range(3):
    step()
torch.cuda.synchronize()
reps = 10
with KernelTimer(names) as kt:
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
for n in names:
    ms, calls, launches = kt.total_ms([n])
    if calls:
        print("%-28s %3d calls/step  %7.1f us per call  (%d launches per call)" % (n, calls // reps, 1e3 * ms / calls, launches // calls))
print("all library kernels: %.1f us per step" % (1e3 * kt.all_ms()[0] / reps))
