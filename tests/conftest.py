import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """Write the parity report of a GPU session (tests/helpers.py: record_parity)."""
    try:
        from tests import helpers
    except Exception:
        return
    if not helpers.PARITY:
        return
    import json

    out = os.environ.get("TDMPC2_PARITY_REPORT", os.path.join(ROOT, "gpurun_out", "parity_r02.json"))
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        with open(out, "w") as f:
            json.dump({"tolerances": {"value_rel": helpers.VALUE_RTOL, "action_abs": helpers.ACT_ATOL},
                       "exit_status": int(exitstatus), "cases": helpers.PARITY}, f, indent=1, sort_keys=True)
    except OSError:
        pass
