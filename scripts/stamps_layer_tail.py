"""Clock stamps inside layer_tail_fwd_kernel (library built with -DLT_STAMPS): where a workgroup's time goes."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import _lib
from athena_amd.layers.commons import bind_all
from athena_amd.layers.transformer import TransformerEncoderLayer
from athena_amd.runtime import ParamStore

dev = torch.device("cuda", 0)
B, T, d, dff = 32, int(os.environ.get("STAMP_T", "250")), 256, 1024
store = ParamStore(torch.bfloat16, device=dev)
layer = TransformerEncoderLayer(store, "enc", d, 4, dff, 0.1)
store.finalize(seed=1)
bind_all(layer)
store.refresh_shadows(force=True)
x = torch.randn(B, T, d, device=dev).to(torch.bfloat16)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 64)()
with torch.no_grad():
    for _ in range(5):
        layer(x, None, training=True)
    torch.cuda.synchronize()
    lib.athena_layer_tail_debug_stamps(buf)
t = list(buf)
names = {0: "start", 1: "prologue done", 2: "ctx Wo", 3: "LN1 + barrier", 4: "z1 y1 out", 5: "LN2", 6: "z2 y2 out"}
for c in range(4):
    names.update({8 + 5 * c: "s%d mul B" % c, 9 + 5 * c: "s%d epilogue" % c, 10 + 5 * c: "s%d barrier" % c,
                  11 + 5 * c: "s%d copy out" % c, 12 + 5 * c: "s%d mul C" % c})
order = [0, 1, 2, 3, 4] + [8 + 5 * c + k for c in range(4) for k in range(5)] + [5, 6]
prev = t[0]
for i in order:
    print("%-16s +%6d cycles   (total %7d)" % (names[i], t[i] - prev, t[i] - t[0]))
    prev = t[i]
