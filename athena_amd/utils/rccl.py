"""A communicator of the solver's own on RCCL, bound through ctypes.

``torch.distributed`` keeps a watchdog thread that polls the completion event of every collective it
issues; an event recorded while a stream is capturing may not be queried, so a collective issued through
the process group cannot safely become a node of a HIP graph.  The data-parallel step records its
gradient all-reduces in its graph (reference: the averaging of ``hvd.DistributedGradientTape``,
athena/solver.py:162-174), so it calls ``ncclAllReduce`` itself, on a stream it chooses, through a
communicator created once per process: rank 0 draws the unique id, the process group carries it to the
other ranks (the same bootstrap torch's own backend performs).
"""
import ctypes
import os

import torch
import torch.distributed as dist

NCCL_FLOAT32 = 7
NCCL_SUM = 0


class _UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_byte * 128)]


def _load():
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    lib = ctypes.CDLL(path)
    lib.ncclGetUniqueId.restype = ctypes.c_int
    lib.ncclGetUniqueId.argtypes = [ctypes.POINTER(_UniqueId)]
    lib.ncclCommInitRank.restype = ctypes.c_int
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, _UniqueId, ctypes.c_int]
    lib.ncclAllReduce.restype = ctypes.c_int
    lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_void_p, ctypes.c_void_p]
    lib.ncclCommDestroy.restype = ctypes.c_int
    lib.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    lib.ncclGetErrorString.restype = ctypes.c_char_p
    lib.ncclGetErrorString.argtypes = [ctypes.c_int]
    return lib


def all_ranks_agree(ok):
    """True when ``ok`` holds on every rank of the default process group (a MIN all-reduce of the flag).
    Every rank must call it at the same point: it is how a decision taken from a local failure
    becomes the same decision everywhere."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return bool(ok)
    on_gpu = dist.get_backend() == "nccl"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda" if on_gpu else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(int(flag.item()))


class Communicator:
    """One RCCL communicator over the ranks of the default process group (one GPU per rank).

    Construction is collective and fails on every rank or on none: a rank on which a local step
    fails (loading the library, drawing the unique id) still takes part in the broadcast of the id and
    in the agreement that follows, so no rank is left waiting inside ``ncclCommInitRank`` for one that
    gave up."""

    def __init__(self):
        if not dist.is_initialized():
            raise RuntimeError("athena_amd.rccl: the process group is not initialised")
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.comm = None
        err = None
        uid = _UniqueId()
        try:
            self.lib = _load()
            if self.rank == 0:
                self._check(self.lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        except (OSError, RuntimeError, AttributeError) as e:
            err = e
        box = [bytes(bytearray(uid.internal)) if (self.rank == 0 and err is None) else None]
        dist.broadcast_object_list(box, src=0)
        if not all_ranks_agree(err is None and box[0] is not None):
            raise RuntimeError("athena_amd.rccl: no communicator (%s)" % (err or "another rank could not prepare one"))
        ctypes.memmove(ctypes.byref(uid), box[0], 128)
        comm = ctypes.c_void_p()
        rc = self.lib.ncclCommInitRank(ctypes.byref(comm), self.world, uid, self.rank)
        if rc == 0:
            self.comm = comm
        if not all_ranks_agree(rc == 0):
            self.close()
            raise RuntimeError("athena_amd.rccl: ncclCommInitRank failed%s" % (
                ": " + self.lib.ncclGetErrorString(rc).decode() if rc != 0 else " on another rank"))

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError("athena_amd.rccl: %s failed: %s" % (what, self.lib.ncclGetErrorString(rc).decode()))

    def all_reduce_sum_(self, t, stream=None):
        """In-place sum of a contiguous fp32 device tensor over the ranks, enqueued on ``stream`` (the
        current stream by default).  Nothing is allocated and nothing waits: the call may be recorded
        in a HIP graph."""
        if t.dtype != torch.float32 or not t.is_cuda or not t.is_contiguous():
            raise RuntimeError("athena_amd.rccl: all_reduce_sum_ takes a contiguous fp32 device tensor")
        s = (stream or torch.cuda.current_stream()).cuda_stream
        self._check(self.lib.ncclAllReduce(t.data_ptr(), t.data_ptr(), t.numel(), NCCL_FLOAT32, NCCL_SUM, self.comm, s),
                    "ncclAllReduce")

    def self_test(self):
        """The communicator against the process group on a small tensor, eagerly and through a recorded
        graph; raises if either disagrees."""
        ref = torch.arange(1024, dtype=torch.float32, device="cuda") * (self.rank + 1)
        want = ref.clone()
        dist.all_reduce(want, op=dist.ReduceOp.SUM)
        got = ref.clone()
        self.all_reduce_sum_(got)
        torch.cuda.synchronize()
        if not torch.equal(got, want):
            raise RuntimeError("athena_amd.rccl: eager all-reduce disagrees with the process group")
        buf = ref.clone()
        side = torch.cuda.Stream()
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            buf.mul_(1.0)
            side.wait_stream(torch.cuda.current_stream())
            self.all_reduce_sum_(buf, side)
            torch.cuda.current_stream().wait_stream(side)
            buf.add_(0.0)
        buf.copy_(ref)
        g.replay()
        torch.cuda.synchronize()
        if not torch.equal(buf, want):
            raise RuntimeError("athena_amd.rccl: recorded all-reduce disagrees with the process group")

    def close(self):
        if getattr(self, "comm", None):
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None
