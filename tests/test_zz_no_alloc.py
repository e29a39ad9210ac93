"""The allocation-free CTC forward entry point against the ordinary one (through the C ABI).

``athena_ctc_loss_fwd_static`` keeps its workspace in a device array inside the library.  Two calls
in a row, the second on a smaller problem, must give the losses and greedy paths of
``athena_ctc_loss_fwd`` bit for bit, and a problem beyond the built-in buffer must be refused.
"""
import ctypes

import numpy as np
import pytest
import torch

from athena_amd import _lib
from tests.util import dev

pytestmark = pytest.mark.gpu


def _case(B, T, V, U, seed):
    rng = np.random.default_rng(seed)
    ldl = (V + 7) // 8 * 8
    logits = torch.zeros(B, T, ldl, dtype=torch.float32)
    logits[..., :V] = torch.from_numpy(np.float32(rng.normal(size=(B, T, V)) * 2.0))
    labels = torch.from_numpy(rng.integers(1, V - 1, size=(B, U)).astype(np.int32))
    llen = torch.from_numpy(rng.integers(max(1, U // 2), U + 1, size=B).astype(np.int32))
    tlen = torch.from_numpy(rng.integers(min(T, 2 * U + 2), T + 1, size=B).astype(np.int32))
    for b in range(B):
        labels[b, int(llen[b]):] = 0
    # one sample whose labels do not fit (the slow path keeps its lattice in the workspace too)
    tlen[B - 1] = max(1, int(llen[B - 1]) - 1)
    return [t.to(dev()) for t in (logits, labels, llen, tlen)], ldl


def _call(lib, static, tensors, ldl, B, T, V, U):
    logits, labels, llen, tlen = tensors
    loss = torch.full((B,), float("nan"), dtype=torch.float32, device=dev())
    best = torch.full((B * T,), -7, dtype=torch.int32, device=dev())
    stream = torch.cuda.current_stream().cuda_stream
    if static:
        rc = lib.athena_ctc_loss_fwd_static(logits.data_ptr(), _lib.F32, ldl, labels.data_ptr(), llen.data_ptr(),
                                            tlen.data_ptr(), B, T, V, U, V - 1, loss.data_ptr(), best.data_ptr(),
                                            stream)
    else:
        n = lib.athena_ctc_workspace_bytes(B, T, U)
        ws = torch.empty(n, dtype=torch.uint8, device=dev())
        rc = lib.athena_ctc_loss_fwd(logits.data_ptr(), _lib.F32, ldl, labels.data_ptr(), llen.data_ptr(),
                                     tlen.data_ptr(), B, T, V, U, V - 1, loss.data_ptr(), best.data_ptr(),
                                     ws.data_ptr(), n, stream)
    torch.cuda.synchronize()
    return rc, loss.cpu(), best.cpu()


def test_static_scratch_ctc_forward_matches_workspace_version():
    lib = _lib.load()
    for (B, T, V, U, seed) in [(8, 120, 300, 30, 1), (3, 40, 50, 9, 2)]:     # second call: smaller problem
        tensors, ldl = _case(B, T, V, U, seed)
        rc0, loss0, best0 = _call(lib, False, tensors, ldl, B, T, V, U)
        rc1, loss1, best1 = _call(lib, True, tensors, ldl, B, T, V, U)
        assert rc0 == 0 and rc1 == 0, lib.athena_last_error()
        assert torch.isfinite(loss0).all()
        assert torch.equal(loss0, loss1), (loss0, loss1)
        assert torch.equal(best0, best1)
        assert float(loss0[B - 1]) > 500.0          # the sample that does not fit


def test_static_scratch_ctc_forward_refuses_oversized_problems():
    lib = _lib.load()
    B, T, V, U = 64, 250, 16, 200            # 57 MB of lattices: beyond the 16 MiB built-in buffer
    assert lib.athena_ctc_workspace_bytes(B, T, U) > (16 << 20)
    dummy = torch.zeros(8, dtype=torch.float32, device=dev())
    rc = lib.athena_ctc_loss_fwd_static(dummy.data_ptr(), _lib.F32, V, dummy.data_ptr(), dummy.data_ptr(),
                                        dummy.data_ptr(), B, T, V, U, V - 1, dummy.data_ptr(), None,
                                        torch.cuda.current_stream().cuda_stream)
    assert rc != 0
    assert b"built-in buffer" in lib.athena_last_error()
