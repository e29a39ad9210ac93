"""Clock stamps of layer_tail_fwd_kernel<true, true> (a library built with -DLT_STAMPS, its path in
ATHENA_STAMPS_LIB): the feed-forward tail followed by the three projections of the next layer."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from athena_amd import _lib
_lib.LIB_PATH = os.environ["ATHENA_STAMPS_LIB"]
from athena_amd import ops
from athena_amd.layers.commons import bind_all
from athena_amd.layers.fused import LN_EPS
from athena_amd.layers.transformer import TransformerEncoderLayer
from athena_amd.runtime import ParamStore

dev = torch.device("cuda", 0)
rows, d, dff = 8000, 256, 1024
store = ParamStore(torch.bfloat16, device=dev)
a = TransformerEncoderLayer(store, "a", d, 4, dff, 0.1)
b = TransformerEncoderLayer(store, "b", d, 4, dff, 0.1)
store.finalize(seed=1)
bind_all(a); bind_all(b)
store.refresh_shadows(force=True)
c = torch.randn(rows, d, device=dev).to(torch.bfloat16)
x = torch.randn(rows, d, device=dev).to(torch.bfloat16)
nq = (b.self_attn.wq, b.self_attn.wk, b.self_attn.wv)
lib = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 64)()
for _ in range(5):
    ops.layer_tail_fwd(c, x, a.self_attn.dense, a.norm1, a.ffn0, a.ffn1, a.norm2, LN_EPS, 0.1, store.rng_state, 1, 2, 3,
                       next_qkv=nq)
torch.cuda.synchronize()
lib.athena_layer_tail_debug_stamps(buf)
t = list(buf)
names = {0: "start", 1: "prologue done", 2: "ctx Wo", 3: "LN1 + barrier", 4: "z1 y1 out", 5: "LN2", 6: "z2 y2 out",
         28: "q", 29: "k", 30: "v", 31: "barrier", 32: "q k v out"}
for s in range(4):
    names.update({8 + 5 * s: "s%d mul B" % s, 9 + 5 * s: "s%d epilogue" % s, 10 + 5 * s: "s%d barrier" % s,
                  11 + 5 * s: "s%d copy out" % s, 12 + 5 * s: "s%d mul C" % s})
order = [0, 1, 2, 3, 4] + [8 + 5 * s + k for s in range(4) for k in range(5)] + [5, 6, 28, 29, 30, 31, 32]
prev = t[0]
for i in order:
    print("%-16s +%6d cycles   (total %7d)" % (names[i], t[i] - prev, t[i] - t[0]))
    prev = t[i]
