"""Repeat the bf16 encoder layer (one-launch tails) forward + backward and compare every parameter gradient
with the first repetition: any difference between repetitions is a race."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ATHENA_LAYER_TAIL"] = "1"
from athena_amd.layers.commons import bind_all
from athena_amd.layers.transformer import TransformerEncoderLayer
from athena_amd.runtime import ParamStore
dev = torch.device("cuda", 0)
dff, B, T = 1024, 4, 64
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
first = None
bad = 0
for rep in range(reps):
    store = ParamStore(torch.bfloat16, device=dev)
    layer = TransformerEncoderLayer(store, "enc", 256, 4, dff, 0.0)
    store.finalize(seed=21)
    bind_all(layer)
    g = torch.Generator().manual_seed(21)
    with torch.no_grad():
        for name, p_ in layer.named_parameters():
            if p_.dim() == 1:
                p_.add_(0.3 * torch.randn(p_.shape, generator=g).to(p_.device))
    store.refresh_shadows(force=True)
    x = torch.randn(B, T, 256, generator=g).to(torch.bfloat16)
    dy = torch.randn(B, T, 256, generator=g).to(torch.bfloat16)
    xg = x.to(dev).requires_grad_(True)
    store.grads.zero_()
    y = layer(xg, None, training=True)
    y.backward(dy.to(dev))
    torch.cuda.synchronize()
    got = {n: p_.grad.detach().double().cpu() for n, p_ in layer.named_parameters()}
    got["y"] = y.detach().double().cpu()
    got["dx"] = xg.grad.double().cpu()
    if first is None:
        first = got
        continue
    for k in got:
        d = float((got[k] - first[k]).abs().max())
        if d != 0.0:
            bad += 1
            print("rep %d: %s differs from the first repetition by %.4g (norm %.4g)" % (rep, k, d, float(first[k].norm())))
    # churn the allocator / caches a little between repetitions
    junk = torch.randn(1 << 22, device=dev)
    del junk
print("repetitions %d, differing tensors %d" % (reps, bad))
