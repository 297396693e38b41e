import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The oracle is torch on the CPU.  A GPU box hands the process 128-256 CPUs and torch then starts as many threads, which
    # makes its oneDNN convolutions at these sizes FOUR TIMES SLOWER than 16 threads do (measured on the MI355X box: the two
    # largest three-step cases 52.9 s with the default 128 threads, 13.4 s with 16, 14.4 s with 32, 25.5 s with 64).  Results do
    # not depend on the count: every parity test hands the oracle its sign decisions.
    try:
        import torch
        torch.set_num_threads(min(16, os.cpu_count() or 1))
    except Exception:                                    # pragma: no cover
        pass


def pytest_collection_modifyitems(config, items):
    """GPU tests never run on a box without a GPU, even if -m is not given."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:                                    # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU on this box")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
