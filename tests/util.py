"""Helpers shared by the GPU parity tests."""
import numpy as np
import torch


def dev():
    return torch.device("cuda", 0)


def rel_err(a, b):
    """max |a-b| / max(|b|, tiny) over the whole tensor (b is the oracle)."""
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    den = max(float(b.abs().max()), 1e-30)
    return float((a - b).abs().max()) / den


def rel_l2(a, b):
    """||a - b||_2 / ||b||_2 over the whole tensor (b is the oracle)."""
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    return float((a - b).norm()) / max(float(b.norm()), 1e-30)


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, "%s: relative error %.3e > %.1e" % (what, e, tol)


def to_cd(x, cd):
    """CPU float64 tensor -> GPU tensor of compute dtype ``cd`` (contiguous)."""
    return x.to(torch.float32).to(dev()).to(cd).contiguous()


def rounded(x, cd):
    """The value the GPU sees after rounding a float64 CPU tensor to ``cd`` (as float64)."""
    return x.to(torch.float32).to(cd).double()


TOL = {torch.float32: 2e-5, torch.bfloat16: 2e-2}
