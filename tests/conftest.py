import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (run on the MI355X box)")
    # Multi-rank GPU tests start their ranks through a fork server that is launched HERE, before any
    # test has touched the GPU: the ranks are then forked from a process that never initialised it.
    import multiprocessing
    from multiprocessing import forkserver
    multiprocessing.get_context("forkserver")
    forkserver.ensure_running()


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
