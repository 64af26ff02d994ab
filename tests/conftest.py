import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "ablation: negative-result variants of libian_ablation.so on a GPU box (run with -m ablation; not part of -m gpu)")
    # the CPU oracles (torch twins) run on the host cores: on a wide host (256 logical CPUs on the MI355X box) torch's
    # default of one thread per logical CPU is catastrophically oversubscribed for these 64x64 convolutions
    # (measured: batch-1 reconstruction 53 ms on 1 thread, 6 s on 256 threads) -- cap it
    try:
        import torch
        torch.set_num_threads(max(1, min(64, os.cpu_count() or 1)))
    except Exception:
        pass


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
