"""Clock stamps inside one small GEMM (needs libathena_timing.so built with the STAMP hooks)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import athena_amd._lib as L
L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libathena_timing.so")
from athena_amd import ops
lib = L.load()
lib.athena_debug_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda")
for M, N, K in [(1088, 256, 256), (8000, 256, 256), (8000, 256, 1024), (8000, 1024, 256)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    for _ in range(5):
        ops.gemm(a, b, c, M, N, K, trans_b=True)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    lib.athena_debug_stamps(buf)
    s = list(buf)
    names = ["entry", "prep", "tile0 in LDS", "loop done", "C parked", "C stored"]
    print("M=%d N=%d K=%d  first block (10 ns units since entry):" % (M, N, K),
          ", ".join("%s +%d" % (names[i], s[i] - s[0]) for i in range(1, 6)),
          "| last block: entry +%d, stored +%d" % (s[8] - s[0], s[9] - s[0]),
          "| issue-done (before store ack) +%d" % (s[6] - s[0]),
          "| pair iterations [loads issued, multiplied, set written, barrier, 2nd multiplied, 2nd barrier]:",
          [[s[10 + 6 * h + j] - s[0] for j in range(6)] for h in range(min(8, (K // 64 + 1) // 2))])
