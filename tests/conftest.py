import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """Measured parity errors of this session (tests/_parity.py) -> gpurun_out/parity_stats.json."""
    try:
        from tests import _parity
        if not _parity.STATS:
            return
        import json
        d = os.path.join(ROOT, "gpurun_out")
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_stats.json"), "w") as f:
            json.dump(_parity.summary(), f, indent=1, sort_keys=True)
    except Exception:      # reporting must never fail a run
        pass
