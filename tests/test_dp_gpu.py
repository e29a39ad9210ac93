"""Two data-parallel ranks on one GPU against the same computation done by hand in one process
(reference athena/solver.py:162-174: hvd.DistributedGradientTape averages the gradients of the ranks
before the clip and the update)."""
import multiprocessing
import os
import socket

import pytest
import torch

from tests import dp_worker
from tests.modelutil import TINY, model_config
from tests.test_model_gpu import ZERO_GRAD, build
from tests.util import rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_two_ranks_pipelined_exchange_matches_hand_averaged_gradients(tmp_path):
    """Each rank records its step as the chain of HIP graphs with the real ``_exchange_async`` between
    the pieces (gloo sums on the host, since both ranks share the one GPU).  The reference trajectory
    is computed here without any exchange code: forward/backward on each rank's shard, gradients
    averaged by hand, one clip + Adam update."""
    from athena_amd import BaseSolver, WarmUpAdam
    world, steps = 2, 3
    case = dict(TINY, mc=model_config(32, 16, 4, 3, 1, 64, rate=0.0), B=4)
    opt_cfg = dict(d_model=32, warmup_steps=3, k=0.05, decay_steps=100000, decay_rate=1.0)
    out_path = str(tmp_path / "rank0.pt")
    port = _free_port()
    ctx = multiprocessing.get_context("forkserver")
    procs = [ctx.Process(target=dp_worker.run_rank, args=(r, world, port, case, "fp32", opt_cfg, steps, out_path))
             for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
    hung = [p for p in procs if p.is_alive()]
    for p in hung:
        p.kill()
    assert not hung, "a rank did not finish"
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    got = torch.load(out_path, weights_only=True)
    assert got["pipelined"], "the multi-rank step must be the pipelined one by default"

    # by hand, in this process
    model, samples, _ = build(case, "fp32")
    st = model.store
    solver = BaseSolver(model, WarmUpAdam(dict(opt_cfg)), config={"clip_norm": 0.5, "enable_tf_function": False})
    shards = [model.prepare_samples(dp_worker.shard_of(samples, r, world)) for r in range(world)]
    losses = []
    for _ in range(steps):
        total = torch.zeros_like(st.grads)
        for r in range(world):
            loss, _m = solver._forward_backward(shards[r])
            total += st.grads
            if r == 0:
                losses.append(float(sum(loss.values())) if isinstance(loss, dict) else float(loss))
        st.grads.copy_(total / world)
        solver._update(1.0)
    torch.cuda.synchronize()
    want = st.params[:st.n_train].double().cpu()
    for a, b in zip(got["losses"], losses):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (got["losses"], losses)
    skip = torch.zeros(st.n_train, dtype=torch.bool)
    for s in st.train_slots:
        if s.name.endswith(ZERO_GRAD):
            skip[s.offset:s.offset + s.numel] = True
    assert rel_err(got["params"].double()[~skip], want[~skip]) <= 1e-4


def test_one_rank_on_rccl_records_the_exchange_in_the_step_graph(tmp_path):
    """With RCCL as the backend the data-parallel step is ONE graph: the pieces of the backward pass with
    the all-reduce of each finished range recorded beside them, through the solver's own communicator
    (utils/rccl.py, whose self-test compares it with the process group eagerly and under replay).  One
    rank is all a one-GPU box can hold on RCCL; its trajectory must be the single-process one."""
    from athena_amd import BaseSolver, WarmUpAdam
    steps = 4
    case = dict(TINY, mc=model_config(32, 16, 4, 3, 1, 64, rate=0.0), B=4)
    opt_cfg = dict(d_model=32, warmup_steps=3, k=0.05, decay_steps=100000, decay_rate=1.0)
    out_path = str(tmp_path / "rccl.pt")
    ctx = multiprocessing.get_context("forkserver")
    p = ctx.Process(target=dp_worker.run_single_rank_rccl, args=(_free_port(), case, "fp32", opt_cfg, steps, out_path))
    p.start()
    p.join(timeout=240)
    if p.is_alive():
        p.kill()
        raise AssertionError("the rank did not finish")
    assert p.exitcode == 0, p.exitcode
    got = torch.load(out_path, weights_only=True)
    assert got["own_communicator"] and got["onegraph"]

    model, samples, _ = build(case, "fp32")
    st = model.store
    solver = BaseSolver(model, WarmUpAdam(dict(opt_cfg)), config={"clip_norm": 0.5, "enable_tf_function": False})
    ds = model.prepare_samples(samples)
    losses = [float(solver.train_step(ds)[0]) for _ in range(steps)]
    torch.cuda.synchronize()
    for a, b in zip(got["losses"], losses):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b)), (got["losses"], losses)
    skip = torch.zeros(st.n_train, dtype=torch.bool)
    for s in st.train_slots:
        if s.name.endswith(ZERO_GRAD):
            skip[s.offset:s.offset + s.numel] = True
    assert rel_err(got["params"].double()[~skip], st.params[:st.n_train].double().cpu()[~skip]) <= 1e-4


def test_bench_launches_its_own_ranks():
    """``python bench.py --gpus 2`` outside a launcher starts two ranks itself (the reference starts
    them with ``horovodrun -np N``, README.md:243-251) and prints ONE line for the whole job.  The two
    ranks share this box's GPU, so the sums go through gloo; the step kinds and the exchange plan are
    the production ones."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(ATHENA_DIST_BACKEND="gloo")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "tiny",
                          "--steps", "3", "--warmup", "3", "--no-cpu-baseline"], env=env, cwd=root,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=240)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2 * out["config"]["per_gpu_batch"]
    assert out["config"]["parallelism"] == "dp2" and out["steps"] == 3
    assert out["value"] > 0 and out["final_loss"] == out["final_loss"]
