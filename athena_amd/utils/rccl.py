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


class Communicator:
    """One RCCL communicator over the ranks of the default process group (one GPU per rank)."""

    def __init__(self):
        if not dist.is_initialized():
            raise RuntimeError("athena_amd.rccl: the process group is not initialised")
        self.lib = _load()
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        uid = _UniqueId()
        if self.rank == 0:
            self._check(self.lib.ncclGetUniqueId(ctypes.byref(uid)), "ncclGetUniqueId")
        box = [bytes(bytearray(uid.internal)) if self.rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        ctypes.memmove(ctypes.byref(uid), box[0], 128)
        self.comm = ctypes.c_void_p()
        self._check(self.lib.ncclCommInitRank(ctypes.byref(self.comm), self.world, uid, self.rank), "ncclCommInitRank")

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
