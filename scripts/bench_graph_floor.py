"""Per-kernel cost inside a replayed HIP graph: a chain of N identical launches, time / N."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from athena_amd import ops
dev = torch.device("cuda")


def chain(fn, n=400, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
            for _ in range(n):
                fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * n)


x = torch.zeros(64, device=dev)
print("torch add_ on 64 floats        %.2f us" % chain(lambda: x.add_(1.0)))
for M, N, K in [(1088, 256, 256), (8000, 256, 64), (8000, 256, 256), (8000, 256, 1024), (8000, 1024, 256), (8000, 768, 256)]:
    a = torch.randn(M, K, device=dev).to(torch.bfloat16)
    b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = chain(lambda: ops.gemm(a, b, c, M, N, K, trans_b=True))
    print("gemm M=%5d N=%5d K=%5d      %.2f us  %.0f TFLOP/s" % (M, N, K, t, 2.0 * M * N * K / t / 1e6))

# does a kernel cost more when its neighbours in the graph are other kernels (cold instruction cache)?
M, N, K = 1088, 256, 256
a = torch.randn(M, K, device=dev).to(torch.bfloat16)
b = torch.randn(N, K, device=dev).to(torch.bfloat16)
bn = torch.randn(K, N, device=dev).to(torch.bfloat16)
c = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
gamma = torch.ones(N, device=dev)
beta = torch.zeros(N, device=dev)
rng = torch.zeros(2, dtype=torch.int64, device=dev)
f_nt = lambda: ops.gemm(a, b, c, M, N, K, trans_b=True)
f_nn = lambda: ops.gemm(a, bn, c, M, N, K, trans_b=False)
f_ln = lambda: ops.add_layernorm_fwd(c, None, gamma, beta, 1e-8, 0.0, rng, 0)
t_nt, t_nn, t_ln = chain(f_nt), chain(f_nn), chain(f_ln)
def mix():
    f_nt(); f_ln(); f_nn(); f_ln()
t_mix = chain(mix, n=100) / 4
print("alone: gemm NT %.2f us, gemm NN %.2f us, layernorm %.2f us -> mean of the mix would be %.2f us; measured in a mixed chain %.2f us"
      % (t_nt, t_nn, t_ln, (t_nt + t_nn + 2 * t_ln) / 4, t_mix))
