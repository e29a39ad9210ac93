"""Cost of cutting the step into several HIP graphs (no communication: the exchange is a no-op)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from oracle import model as om
wl = bench.WORKLOADS["aishell_b32_t1000"]
for mode in ("single", "split2", "pipelined"):
    os.environ["ATHENA_DP_PIPELINE"] = "1" if mode == "pipelined" else "0"
    from athena_amd import BaseSolver, HorovodSolver, MtlTransformerCtc, WarmUpAdam
    from tests.modelutil import DataDescriptions, model_config
    mc = model_config(wl["d_model"], wl["num_filters"], wl["heads"], wl["enc"], wl["dec"], wl["dff"], wl["rate"])
    model = MtlTransformerCtc(DataDescriptions(wl["vocab"], wl["feat"]), mc, precision="bf16", seed=1)
    opt = WarmUpAdam(dict(wl["opt"]))
    cfg = {"clip_norm": 100.0, "log_interval": 10 ** 9, "enable_tf_function": True}
    if mode == "single":
        solver = BaseSolver(model, opt, config=cfg)
    else:
        solver = HorovodSolver(model, opt, config=cfg)
        solver.world_size = 2
        solver._exchange_async = lambda t: None
    ds = model.prepare_samples(om.make_samples(wl["B"], wl["T"], wl["feat"], wl["vocab"] + 1, wl["U"], seed=1))
    for _ in range(5):
        solver.train_step(ds)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        solver.train_step(ds)
    torch.cuda.synchronize()
    print("%-10s %.3f ms/step" % (mode, (time.perf_counter() - t0) / 20 * 1e3))
    del solver, model, opt
    torch.cuda.empty_cache()
