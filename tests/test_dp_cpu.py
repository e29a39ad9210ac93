"""The data-parallel path on two CPU processes (torch.distributed, gloo): the bucketed all-reduce
of the flat gradient buffer gives the mean over ranks once the optimizer's 1/world_size is
applied, every bucket boundary is a variable boundary, and broadcast_variables makes the ranks
agree.  On the GPUs the same code runs over RCCL (backend "nccl")."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Store:
    """The part of ParamStore the solver touches, on the CPU."""

    def __init__(self, rank):
        sizes = [64, 128, 1024, 64, 4096, 64, 4096, 64]
        offs, o = [], 0
        for s in sizes:
            offs.append(o)
            o += s
        self.n_train = o
        self.device = torch.device("cpu")
        self.train_slots = list(range(len(sizes)))
        self.seg_offsets = torch.tensor(offs + [o], dtype=torch.int64)
        g = torch.Generator().manual_seed(100 + rank)
        self.params = torch.randn(o + 32, generator=g)            # tail: non-trainable statistics
        self.grads = torch.randn(o, generator=g)
        self.shadows_dirty = False
        self.rng_state = torch.zeros(2, dtype=torch.int64)


class _Model:
    def __init__(self, rank):
        self.store = _Store(rank)
        self.metric = None


def _worker(rank, world, port, bucket_mb, ret):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from athena_amd import HorovodSolver, WarmUpAdam
    HorovodSolver.BUCKET_MB = HorovodSolver.EXCHANGE_MB = bucket_mb
    HorovodSolver.initialize_devices()
    assert dist.is_initialized() and dist.get_backend() == "gloo"
    model = _Model(rank)
    solver = HorovodSolver(model, WarmUpAdam({"d_model": 32}), config={"enable_tf_function": False})
    assert solver.world_size == world and solver.rank == rank
    mine = model.store.grads.clone()
    scale = solver._reduce_gradients()
    both = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    want = sum(both) / world
    ok_mean = torch.allclose(model.store.grads * scale, want, atol=1e-6)
    # buckets: cover [0, n) from the end, cut on variable boundaries
    offs = set(model.store.seg_offsets.tolist())
    b = solver._buckets
    ok_buckets = b[0][1] == model.store.n_train and b[-1][0] == 0 and all(x in offs and y in offs for x, y in b)
    # broadcast: parameters and optimizer slots of rank 0 everywhere
    solver.optimizer.m.fill_(float(rank + 1))
    solver.broadcast_variables()
    p0 = [torch.zeros_like(model.store.params) for _ in range(world)]
    dist.all_gather(p0, model.store.params)
    ok_bcast = all(torch.equal(p0[0], t) for t in p0) and float(solver.optimizer.m[0]) == 1.0 and \
        model.store.shadows_dirty
    # a decision taken from a local failure becomes everybody's: one rank saying no is a no everywhere
    from athena_amd.utils.rccl import all_ranks_agree
    ok_agree = all_ranks_agree(True) and not all_ranks_agree(rank != 1) and solver._rccl is None
    ret[rank] = (ok_mean, ok_buckets, ok_bcast and ok_agree, len(b), scale)
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("bucket_mb", [25.0, 0.004])
def test_gradient_average_over_two_ranks(bucket_mb):
    world = 2
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_mb, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0, "rank failed (exit code %s)" % p.exitcode
    for r in range(world):
        ok_mean, ok_buckets, ok_bcast, nb, scale = ret[r]
        assert ok_mean and ok_buckets and ok_bcast
        assert scale == 0.5
        assert nb == 1 if bucket_mb > 1 else nb > 3
