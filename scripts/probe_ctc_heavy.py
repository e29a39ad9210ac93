"""CTC launches at the CTC-heavy shape (B 64, 250 frames, V 8001, U 200): five passes without and five with the\ngradient, for a kernel trace."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops
dev = torch.device("cuda", 0)
B, T, V, U = 64, 250, 8001, 200
ldl = ops.roundup(V, 8)
logits = torch.randn(B, T, ldl, device=dev)
labels = torch.randint(1, V - 1, (B, U), device=dev, dtype=torch.int32)
ll = torch.full((B,), U, dtype=torch.int32, device=dev)
tl = torch.full((B,), T, dtype=torch.int32, device=dev)
ws = torch.empty(ops.ctc_workspace_bytes(B, T, U), dtype=torch.uint8, device=dev)
per = torch.empty(B, device=dev)
g16 = torch.empty(B, T, ldl, device=dev, dtype=torch.bfloat16)
for _ in range(5):
    ops.ctc_loss_fwd(logits, ldl, labels, ll, tl, B, T, V, U, V - 1, per, ws)
torch.cuda.synchronize()
for _ in range(5):
    ops.ctc_loss_fwd_bwd(logits, ldl, labels, ll, tl, B, T, V, U, V - 1, per, ws, 1.0 / B, g16)
torch.cuda.synchronize()
print("done")
