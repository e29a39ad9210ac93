"""Throughput of the hybrid attention/CTC training step on MI355X.

python bench.py --gpus N --steps K --warmup W     (N > 1: under torch.distributed.run one rank of N;
                                                    outside a launcher it starts its own N ranks)

One step = forward + backward + per-tensor clip + Adam of MtlTransformerCtc on one synthetic
batch (per-rank batch fixed, so N ranks process N batches per step: weak scaling).  Prints ONE
JSON line on rank 0: audio-seconds per second over all ranks, plus the roofline figures of the
dominant kernel (measured with HIP events on the launch stream) and the CPU oracle's rate.
"""
import argparse
import copy
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FRAME_SHIFT_S = 0.010      # 10 ms fbank frame shift

WORKLOADS = {
    # AISHELL-sized MTL Transformer-CTC: 12 enc / 6 dec, d=256, 4 heads, 4233 classes, batch 32, T=1000, 80-dim
    "aishell_b32_t1000": dict(B=32, T=1000, feat=80, vocab=4232, U=32, d_model=256, num_filters=256, heads=4,
                              enc=12, dec=6, dff=1024, rate=0.1, opt=dict(d_model=256, warmup_steps=25000, k=1.0,
                                                                          decay_steps=100000000, decay_rate=0.1),
                              clip_norm=100.0),
    # TIMIT recipe (the reference's CPU-runnable case)
    "timit_b16_t500": dict(B=16, T=500, feat=40, vocab=49, U=40, d_model=256, num_filters=256, heads=8, enc=9,
                           dec=3, dff=1024, rate=0.2, opt=dict(d_model=256, warmup_steps=4000, k=1.0),
                           clip_norm=100.0),
    # LibriSpeech recipe dimensions on long utterances (attention over 750 frames, long CTC lattice)
    "librispeech_b8_t3000": dict(B=8, T=3000, feat=80, vocab=5000, U=150, d_model=512, num_filters=512, heads=8,
                                 enc=12, dec=6, dff=2048, rate=0.1,
                                 opt=dict(d_model=512, warmup_steps=25000, k=1.0), clip_norm=5.0),
    # AISHELL recipe as shipped (d=512, 8 heads, 40-dim features)
    "aishell_shipped_b32_t1000": dict(B=32, T=1000, feat=40, vocab=4232, U=32, d_model=512, num_filters=512,
                                      heads=8, enc=12, dec=6, dff=1280, rate=0.1,
                                      opt=dict(d_model=512, warmup_steps=25000, k=1.0), clip_norm=100.0),
    # CTC-heavy stress: encoder + CTC head only carry gradient (no decoder layers, mtl_weight 0: the loss is
    # the CTC term alone), 8k classes, batch 64, 200 labels per utterance
    "ctc_heavy_b64": dict(B=64, T=1000, feat=80, vocab=8000, U=200, d_model=256, num_filters=256, heads=4, enc=12,
                          dec=0, dff=1024, rate=0.1, mtl_weight=0.0,
                          opt=dict(d_model=256, warmup_steps=25000, k=1.0), clip_norm=100.0),
    "tiny": dict(B=4, T=64, feat=16, vocab=30, U=5, d_model=64, num_filters=32, heads=2, enc=2, dec=1, dff=128,
                 rate=0.1, opt=dict(d_model=64, warmup_steps=10, k=1.0), clip_norm=100.0),
}

HBM_PEAK_GBS = 8000.0       # MI355X HBM3E peak
MFMA_BF16_PEAK_TFLOPS = 2500.0


def build(wl, precision, seed):
    from athena_amd import BaseSolver, HorovodSolver, MtlTransformerCtc, WarmUpAdam
    from athena_amd.data import DataDescriptions, model_config
    mc = model_config(wl["d_model"], wl["num_filters"], wl["heads"], wl["enc"], wl["dec"], wl["dff"], wl["rate"],
                      mtl_weight=wl.get("mtl_weight", 0.5))
    model = MtlTransformerCtc(DataDescriptions(wl["vocab"], wl["feat"]), mc, precision=precision, seed=seed)
    opt = WarmUpAdam(dict(wl["opt"]))
    cfg = {"clip_norm": wl["clip_norm"], "log_interval": 10 ** 9, "enable_tf_function": True}
    solver = (HorovodSolver if dist.is_initialized() else BaseSolver)(model, opt, config=cfg)
    return model, opt, solver, mc


def gemm_flops_parts(wl):
    """Algorithmic FLOPs of one training step by GEMM family: the implicit-GEMM second convolution
    (forward, input gradient, weight gradient), the forward and input-gradient products of the
    vocabulary projections, those of every other Dense layer (x_net Dense, q/k/v/o projections and
    FFN of the encoder and decoder layers), and the weight gradients of all Dense layers."""
    B, T, Fd = wl["B"], wl["T"], wl["feat"]
    C, d, dff, V, U = wl["num_filters"], wl["d_model"], wl["dff"], wl["vocab"] + 1, wl["U"]
    T1, F1 = (T + 1) // 2, (Fd + 1) // 2
    T2, F2 = (T1 + 1) // 2, (F1 + 1) // 2
    L = U + 1
    conv2 = 2 * 9 * C * C * T2 * F2
    xdense = 2 * F2 * C * d * T2
    enc = wl["enc"] * (8 * T2 * d * d + 4 * T2 * d * dff)
    dec = wl["dec"] * (8 * L * d * d + 4 * L * d * d + 4 * T2 * d * d + 4 * L * d * dff)
    heads = 2 * L * d * V + 2 * T2 * d * V
    return {"conv2": 3.0 * B * conv2, "layer_gemm": 2.0 * B * (xdense + enc + dec), "vocab": 2.0 * B * heads,
            "dense_wgrad": 1.0 * B * (xdense + enc + dec + heads)}


def gemm_flops_per_step(wl):
    """Algorithmic FLOPs of all Dense/Conv2D contractions of one training step (forward x 3)."""
    B, T, Fd = wl["B"], wl["T"], wl["feat"]
    C, d, dff, V, U = wl["num_filters"], wl["d_model"], wl["dff"], wl["vocab"] + 1, wl["U"]
    T1, F1 = (T + 1) // 2, (Fd + 1) // 2
    T2, F2 = (T1 + 1) // 2, (F1 + 1) // 2
    L = U + 1
    conv2 = 2 * 9 * C * C * T2 * F2
    xdense = 2 * F2 * C * d * T2
    enc = wl["enc"] * (8 * T2 * d * d + 4 * T2 * d * dff)
    dec = wl["dec"] * (8 * L * d * d + 4 * L * d * d + 4 * T2 * d * d + 4 * L * d * dff)
    heads = 2 * L * d * V + 2 * T2 * d * V
    return 3.0 * B * (conv2 + xdense + enc + dec + heads)


def ctc_bytes(wl):
    """Algorithmic HBM bytes of the CTC loss per step: fp32 logits read twice, the gradient written
    once (bf16 in the bf16 precision mode: the vocabulary projection's backward pass consumes it
    as is), the two lattices written and read."""
    T2 = (((wl["T"] + 1) // 2) + 1) // 2
    V = wl["vocab"] + 1
    S = 2 * wl["U"] + 1
    grad_bytes = 2 if wl.get("precision", "bf16") == "bf16" else 4
    return wl["B"] * T2 * V * (4 * 2 + grad_bytes) + wl["B"] * T2 * S * 4 * 2


def attention_flops_per_step(wl):
    """QK^T and PV of every attention block (forward x 3.5: the backward recomputes the scores)."""
    B, d, U = wl["B"], wl["d_model"], wl["U"]
    T2 = (((wl["T"] + 1) // 2) + 1) // 2
    L = U + 1
    fwd = wl["enc"] * 4 * T2 * T2 * d + wl["dec"] * (4 * L * L * d + 4 * L * T2 * d)
    return 3.5 * B * fwd


class KernelTimer:
    """Per-kernel durations of selected C-ABI entry points from the library's timing sessions.

    While a session is open (``athena_timing_begin``) every kernel the library launches carries a pair
    of events that the runtime fills with the kernel's own begin and end timestamps -- the figures a
    profiler reports, free of launch gaps and of the cost of the events.  The entry points are wrapped
    at the ctypes level: the launches an entry point makes are the ones between the session's launch
    count before and after the call.  ``classify(name, args)`` names the bucket a call is booked under."""

    def __init__(self, names, classify=None):
        from athena_amd import _lib
        self.lib = _lib.load()
        self.names = names
        self.classify = classify or (lambda name, args: name)
        self.ranges = {}          # bucket -> [(first launch, launches)]
        self._orig = {}
        self.ms = None

    def __enter__(self):
        for n in self.names:
            fn = getattr(self.lib, n)
            self._orig[n] = fn

            def wrapped(*a, _fn=fn, _n=n):
                n0 = self.lib.athena_timing_count()
                r = _fn(*a)
                n1 = self.lib.athena_timing_count()
                if n1 > n0:
                    self.ranges.setdefault(self.classify(_n, a), []).append((n0, n1 - n0))
                return r
            setattr(self.lib, n, wrapped)
        self.lib.athena_timing_begin()
        return self

    def __exit__(self, *exc):
        import ctypes
        total = self.lib.athena_timing_end()
        for n, fn in self._orig.items():
            setattr(self.lib, n, fn)
        torch.cuda.synchronize()
        buf = (ctypes.c_float * max(total, 1))()
        if total and self.lib.athena_timing_read(0, total, buf) != 0:
            raise RuntimeError("athena_timing_read: %s" % self.lib.athena_last_error().decode())
        self.ms = list(buf)[:total]

    def total_ms(self, keys):
        """(milliseconds of kernel time, entry-point calls, kernel launches) of the buckets ``keys``."""
        calls = [r for k in keys for r in self.ranges.get(k, ())]
        return (sum(sum(self.ms[a:a + n]) for a, n in calls), len(calls), sum(n for _, n in calls))

    def all_ms(self):
        """Kernel time and launches of everything the library launched in the session."""
        return sum(self.ms), len(self.ms)


GEMM_ENTRIES = ["athena_gemm", "athena_gemm_grouped", "athena_gemm_wgrad_table", "athena_layer_tail_fwd",
                "athena_layer_tail_bwd", "athena_panel_gemm", "athena_panel_gemm_k", "athena_conv3x3s2_fwd",
                "athena_conv3x3s2_dgrad", "athena_conv3x3s2_wgrad"]
GEMM_FAMILIES = ("conv2", "layer_gemm", "vocab", "dense_wgrad")
ATTN_ENTRIES = ["athena_mha_fwd", "athena_mha_bwd"]
CTC_ENTRIES = ["athena_ctc_loss_fwd", "athena_ctc_loss_bwd", "athena_ctc_loss_fwd_bwd", "athena_rescale_unless"]


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(wl, model, mc, seconds_budget=25.0):
    """The CPU oracle (torch-CPU fp32 restatement of the same step) on a bounded sample: a first
    step on 4 utterances sizes the batch that fits the time budget (the whole batch when it does)."""
    from oracle import model as om
    V = wl["vocab"] + 1
    threads = min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16") or 16))
    torch.set_num_threads(threads)
    ocfg = dict(mc["model_config"], num_class=V, feat_dim=wl["feat"], mtl_weight=mc["mtl_weight"])
    state = model.store.state_dict()

    def run(Bs, steps):
        p = {k: v.clone() for k, v in state.items()}
        for k, v in p.items():
            if not k.endswith(om.TRAINABLE_SKIP):
                v.requires_grad_(True)
        smp = om.make_samples(Bs, wl["T"], wl["feat"], V, wl["U"], seed=1, full_length=True)
        st = {"iterations": 0, "m": {}, "v": {}}
        times = []
        for _ in range(steps):
            t0 = time.time()
            om.train_step(p, st, smp, ocfg, {"clip_norm": wl["clip_norm"]}, dict(wl["opt"]))
            times.append(time.time() - t0)
        return times

    t_all = time.time()
    probe = min(wl["B"], 4)
    first = run(probe, 2)
    per_utt = first[-1] / probe
    # two more steps of the largest batch (a multiple of the probe) that fits what is left of the budget
    left = seconds_budget - (time.time() - t_all)
    Bs = probe
    for cand in sorted({probe * 2 ** i for i in range(1, 8)} | {wl["B"]}):
        if cand <= wl["B"] and per_utt * cand * 2 <= left:
            Bs = max(Bs, cand)
    times = run(Bs, 2) if Bs > probe else first
    t = min(times[1:]) if len(times) > 1 else times[0]
    return {"value": Bs * wl["T"] * FRAME_SHIFT_S / t, "unit": "audio-sec/sec", "cores": threads, "kind": "port",
            "host_cpu_count": os.cpu_count(), "cpu_model": _cpu_model(),
            "sample": "%d of %d utterances per step (T=%d), 2 steps, the second one timed; torch-CPU fp32 "
                      "oracle of the same training step on %d threads" % (Bs, wl["B"], wl["T"], threads)}


def pmc_traffic(workload, family):
    """HBM bytes per launch of a kernel family from the committed PMC passes (FETCH_SIZE doubled as
    the gfx950 counter reports half of a wide streaming read, plus WRITE_SIZE; made by
    scripts/pmc_traffic.py from two ``rocprofv3 --pmc`` runs of this benchmark), or None.  Counters
    cannot be read from inside the benchmark: this one figure of the line comes from a file."""
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ("r3_pmc_traffic.json", "r2_pmc_traffic.json"):
        try:
            with open(os.path.join(here, "profiles", name)) as f:
                d = json.load(f)
            if d.get("workload") != workload:
                continue
            return d["families"][family]["hbm_bytes_per_launch"]
        except (OSError, KeyError, ValueError):
            continue
    return None


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(n):
    """``python bench.py --gpus N`` outside a launcher: start the N ranks as child processes (one per
    GPU, the environment torch.distributed.run would give them), relay rank 0's JSON line, and fail if
    any rank fails.  Nothing here touches the GPU, the parent only waits."""
    import subprocess
    port = str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    line = procs[0].stdout.read().decode()
    codes = [pr.wait() for pr in procs]
    if any(codes):
        sys.stderr.write("bench.py: rank exit codes %s\n" % codes)
        return 1
    sys.stdout.write(line)
    sys.stdout.flush()
    return 0


def logit_error_of_bf16(wl, samples):
    """What the bf16 mode does to the logits of THIS workload's batch: the same random-init weights
    (same seed) run forward once in the library's fp32 mode (the mode the parity tests hold to 5e-5
    of the fp64 CPU oracle) and once in bf16 mode, dropout off, batch statistics in BatchNorm as in a
    training step.  Root-mean-square and largest absolute difference per head."""
    from athena_amd import MtlTransformerCtc
    from athena_amd.data import DataDescriptions, model_config
    mc = model_config(wl["d_model"], wl["num_filters"], wl["heads"], wl["enc"], wl["dec"], wl["dff"], 0.0,
                      mtl_weight=wl.get("mtl_weight", 0.5))
    outs = {}
    for prec in ("fp32", "bf16"):
        m = MtlTransformerCtc(DataDescriptions(wl["vocab"], wl["feat"]), copy.deepcopy(mc), precision=prec, seed=1)
        with torch.no_grad():
            att, ctc = m(m.prepare_samples(samples), training=True)
        outs[prec] = (att.float().cpu(), ctc.float().cpu())
        del m, att, ctc
    torch.cuda.empty_cache()
    rms, mx, scale = [], [], 0.0
    for a, b in zip(outs["fp32"], outs["bf16"]):
        d = (a - b).double()
        rms.append(float(d.pow(2).mean().sqrt()))
        mx.append(float(d.abs().max()))
        scale = max(scale, float(a.abs().max()))
    return {"rms_abs": rms, "max_abs": mx, "logit_scale": scale, "heads": ["attention", "ctc"],
            "target_abs": 1e-2, "target_met": ("max" if max(mx) <= 1e-2 else "rms only" if max(rms) <= 1e-2 else "no"),
            "measured": "forward of this batch in fp32 mode vs bf16 mode of the library, same weights, dropout off"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="aishell_b32_t1000")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-logit-error", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: this process becomes the launcher of its own N ranks
        sys.exit(launch_ranks(args.gpus))

    # stdout carries the one JSON line and nothing else: libraries that write banners to file
    # descriptor 1 (RCCL prints its version block there at initialisation) are sent to stderr, and
    # the line is written to the saved descriptor and flushed before the process group is torn down
    # (a buffered line was lost at exit when stdout was a file)
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ATHENA_DIST_BACKEND=gloo rehearses the multi-rank control flow on a box with fewer GPUs than
    # ranks (ranks then share devices; RCCL itself refuses two ranks on one GPU)
    backend = os.environ.get("ATHENA_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    if world > 1 or os.environ.get("ATHENA_FORCE_DP", "0") == "1":
        # a forced single-process data-parallel run outside a launcher rendezvouses with itself
        for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(key, val)
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend=backend)
    wl = dict(WORKLOADS[args.workload], precision=args.precision)

    wd = float(os.environ.get("ATHENA_BENCH_WATCHDOG", "0"))
    if wd > 0:                       # Python stacks of every thread if the run is still going after wd seconds
        import faulthandler
        faulthandler.dump_traceback_later(wd, exit=False, file=sys.stderr)

    def log(msg):
        sys.stderr.write("[bench rank %d/%d] %s\n" % (rank, world, msg))
        sys.stderr.flush()

    from athena_amd.data import make_samples
    V = wl["vocab"] + 1
    samples = make_samples(wl["B"], wl["T"], wl["feat"], V, wl["U"], seed=1 + rank, full_length=True)
    logit_err = None
    if rank == 0 and world == 1 and args.precision == "bf16" and not args.no_logit_error:
        logit_err = logit_error_of_bf16(wl, samples)
        log("bf16 logit error measured: rms %s max %s" % (logit_err["rms_abs"], logit_err["max_abs"]))
    model, opt, solver, mc = build(wl, args.precision, seed=1)
    if args.no_graph:
        solver.hparams.enable_tf_function = False
    ds = model.prepare_samples(samples)          # resident in HBM before the timed region
    if dist.is_initialized():
        solver.broadcast_variables()

    def barrier():
        torch.cuda.synchronize()
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize()

    log("model built, %d warm-up steps" % args.warmup)
    for i in range(args.warmup):
        loss, metrics = solver.train_step(ds)
        if world > 1:
            torch.cuda.synchronize()
            log("warm-up step %d done" % i)
    model.reset_metrics()
    barrier()
    log("timing %d steps" % args.steps)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, metrics = solver.train_step(ds)
    barrier()
    dt = time.perf_counter() - t0
    log("timed region done: %.3f ms per step" % (dt / args.steps * 1e3))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.detach())
    # the accuracies every timed step scored on the device (two integers each, read here once)
    scored = {k: float(v) for k, v in (metrics or {}).items()}

    # per-kernel timing pass (eager steps inside a timing session of the library).  Only the one-GPU
    # run carries it: with several ranks the others would sit in a collective for its whole length.
    roof = ctc_roof = attn_roof = parts = None
    kernel_ms_all = launches_all = None
    if rank == 0 and world == 1:
        was = solver.hparams.enable_tf_function
        solver.hparams.enable_tf_function = False
        was_dp = (solver.world_size, getattr(solver, "_force_exchange", False))
        solver.world_size, solver._force_exchange = 1, False
        Vp = (V + 7) // 8 * 8

        def classify(name, a):
            if name in ATTN_ENTRIES:
                return "attention"
            if name in CTC_ENTRIES:
                return "ctc"
            if name.startswith("athena_conv3x3s2"):
                return "conv2"
            if name == "athena_gemm_wgrad_table":
                return "dense_wgrad"
            if name in ("athena_gemm", "athena_gemm_grouped"):
                d = a[0]._obj if hasattr(a[0], "_obj") else a[0]
                if name == "athena_gemm_grouped":
                    d = d[0]
                if d.trans_a:
                    return "dense_wgrad"               # a weight gradient issued on its own (x^T dy)
                if name == "athena_gemm" and any(x in (V, Vp) for x in (d.M, d.N, d.K)):
                    return "vocab"
            return "layer_gemm"                     # Dense forward / input gradient, fused layer tails
        nrep = 3
        solver.train_step(ds)
        torch.cuda.synchronize()
        with KernelTimer(GEMM_ENTRIES + ATTN_ENTRIES + CTC_ENTRIES, classify) as kt:
            for _ in range(nrep):
                solver.train_step(ds)
            torch.cuda.synchronize()
        solver.hparams.enable_tf_function = was
        solver.world_size, solver._force_exchange = was_dp
        kernel_ms_all, launches_all = kt.all_ms()
        kernel_ms_all, launches_all = kernel_ms_all / nrep, launches_all // nrep
        fl = gemm_flops_parts(wl)
        g_ms, g_calls, g_launches = kt.total_ms(list(GEMM_FAMILIES))
        flops = sum(fl.values())
        ach = flops * nrep / (g_ms * 1e-3) / 1e12
        roof = {"kernel": "gemm_mfma_kernel / gemm_mfma_grouped_kernel / gemm_big_kernel / gemm_big_table_kernel / "
                          "layer_tail_*_kernel / panel_*_kernel (+ splitk_reduce): every Dense and implicit-GEMM Conv2D "
                          "launch of a step",
                "bound": "mfma", "achieved": ach, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / MFMA_BF16_PEAK_TFLOPS, "traffic": pmc_traffic(args.workload, "gemm"),
                "launches_per_step": g_launches // nrep, "entry_calls_per_step": g_calls // nrep,
                "ms_per_step": g_ms / nrep, "avg_launch_us": 1e3 * g_ms / max(g_launches, 1),
                "timing": "kernel begin/end timestamps of every launch (athena_timing_*), %d eager steps" % nrep,
                "flops_per_step": flops}
        parts = {}
        for key in GEMM_FAMILIES:
            ms, calls, launches = kt.total_ms([key])
            if calls == 0:
                continue
            a_ = fl[key] * nrep / (ms * 1e-3) / 1e12
            parts[key] = {"bound": "mfma", "achieved": a_, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                          "frac": a_ / MFMA_BF16_PEAK_TFLOPS, "ms_per_step": ms / nrep,
                          "launches_per_step": launches // nrep, "flops_per_step": fl[key]}
        a_ms, a_calls, a_launches = kt.total_ms(["attention"])
        if a_calls:
            aa = attention_flops_per_step(wl) * nrep / (a_ms * 1e-3) / 1e12
            attn_roof = {"kernel": "mha_fwd_mfma_kernel, mha_bwd_mfma_kernel", "bound": "mfma",
                         "achieved": aa, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": aa / MFMA_BF16_PEAK_TFLOPS, "traffic": pmc_traffic(args.workload, "attention"),
                         "ms_per_step": a_ms / nrep, "launches_per_step": a_launches // nrep}
        c_ms, _, c_launches = kt.total_ms(["ctc"])
        cb = ctc_bytes(wl) * nrep / (c_ms * 1e-3) / 1e9
        ctc_roof = {"kernel": "ctc_row_kernel (logsumexp, emissions, dense gradient) + ctc_lattice_kernel + ctc_post_kernel "
                              "(+ rescale_unless_kernel in the backward pass)", "bound": "hbm",
                    "achieved": cb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": cb / HBM_PEAK_GBS,
                    "traffic": pmc_traffic(args.workload, "ctc"),
                    "ms_per_step": c_ms / nrep, "launches_per_step": c_launches // nrep,
                    "bytes_per_step": ctc_bytes(wl)}

    if rank == 0:
        audio_s = wl["B"] * wl["T"] * FRAME_SHIFT_S * world * args.steps
        out = {
            "metric": "audio-sec/sec fwd+bwd, Transformer-CTC AISHELL b32 T=1000" if args.workload == "aishell_b32_t1000"
            else "audio-sec/sec fwd+bwd, Transformer-CTC workload %s" % args.workload,
            "value": audio_s / dt, "unit": "audio-sec/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": args.workload, "per_gpu_batch": wl["B"], "global_batch": wl["B"] * world,
                       "frames": wl["T"], "fbank_dim": wl["feat"], "classes": V, "parallelism": "dp%d" % world,
                       "hip_graph": bool(solver.hparams.enable_tf_function), "dropout": wl["rate"],
                       "encoder_layers": wl["enc"], "decoder_layers": wl["dec"],
                       "mtl_weight": wl.get("mtl_weight", 0.5)},
            "rccl_ranks": (solver._rccl.world if getattr(solver, "_rccl", None) is not None else
                           (dist.get_world_size() if dist.is_initialized() and dist.get_backend() == "nccl" else 0)),
            "step_kind": type(next((v for v in getattr(solver, "_graphs", {}).values() if v), None)).__name__,
            "final_loss": final_loss,
            # both accuracies are scored on the device inside every step (arg-max, greedy collapse / cut at
            # eos, edit distance, two integer sums); the values below are those sums over the timed steps
            "metrics_scored_in_timed_region": True, "metrics_over_timed_steps": scored,
            "kernel_ms_per_step_all_launches": kernel_ms_all, "library_launches_per_step": launches_all,
            "roofline": roof, "roofline_parts": parts, "roofline_attention": attn_roof, "roofline_ctc": ctc_roof,
        }
        if logit_err is not None:
            out["bf16_logit_error"] = logit_err
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(wl, model, mc)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
