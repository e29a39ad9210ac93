"""One rank of the two-rank data-parallel rehearsal (started by tests/test_dp_gpu.py through a fork
server): both ranks share GPU 0, the gradient sums go through gloo on the host, everything else is
the production path -- HorovodSolver, the step recorded as a chain of HIP graphs, the exchange of
each piece through ``HorovodSolver._exchange_async`` while the next piece runs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def shard_of(samples, rank, world):
    return {k: v[rank::world].contiguous() for k, v in samples.items()}


def run_rank(rank, world, port, case, precision, opt_cfg, steps, out_path):
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), ATHENA_DP_BACKEND="gloo")
    os.environ.pop("ATHENA_DP_PIPELINE", None)
    import torch
    import torch.distributed as dist
    from athena_amd import HorovodSolver, WarmUpAdam
    from athena_amd.solver import _PipelinedGraphStep
    from tests.test_model_gpu import build
    HorovodSolver.initialize_devices(solver_gpus=[0] * world)
    assert dist.get_backend() == "gloo" and dist.get_world_size() == world
    model, samples, _ = build(case, precision)
    if rank != 0:
        # the broadcast after the first step must bring every rank to rank 0's values
        with torch.no_grad():
            model.store.params[:model.store.n_train].add_(0.01)
    solver = HorovodSolver(model, WarmUpAdam(dict(opt_cfg)),
                           config={"clip_norm": 0.5, "log_interval": 10 ** 9, "enable_tf_function": True})
    solver.broadcast_variables()
    mine = model.prepare_samples(shard_of(samples, rank, world))
    losses = [float(solver.train_step(mine)[0]) for _ in range(steps)]
    torch.cuda.synchronize()
    pipelined = any(isinstance(v, _PipelinedGraphStep) for v in solver._graphs.values())
    if rank == 0:
        torch.save({"params": model.store.params[:model.store.n_train].cpu(), "losses": losses,
                    "pipelined": pipelined}, out_path)
    dist.barrier()
    dist.destroy_process_group()


def run_single_rank_rccl(port, case, precision, opt_cfg, steps, out_path):
    """One rank on RCCL with the exchange forced: the solver's own communicator (checked against the
    process group by its self-test), the step recorded as ONE graph with the all-reduces inside."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      ATHENA_FORCE_DP="1")
    for k in ("ATHENA_DP_PIPELINE", "ATHENA_DP_MODE", "ATHENA_DP_BACKEND"):
        os.environ.pop(k, None)
    import torch
    import torch.distributed as dist
    from athena_amd import HorovodSolver, WarmUpAdam
    from athena_amd.solver import _OneGraphPipelinedStep
    from tests.test_model_gpu import build
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    model, samples, _ = build(case, precision)
    solver = HorovodSolver(model, WarmUpAdam(dict(opt_cfg)),
                           config={"clip_norm": 0.5, "log_interval": 10 ** 9, "enable_tf_function": True})
    mine = model.prepare_samples(samples)
    losses = [float(solver.train_step(mine)[0]) for _ in range(steps)]
    torch.cuda.synchronize()
    torch.save({"params": model.store.params[:model.store.n_train].cpu(), "losses": losses,
                "onegraph": any(isinstance(v, _OneGraphPipelinedStep) for v in solver._graphs.values()),
                "own_communicator": solver._rccl is not None}, out_path)
    dist.destroy_process_group()
