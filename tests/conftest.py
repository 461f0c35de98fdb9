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

    out = os.environ.get("TDMPC2_PARITY_REPORT", os.path.join(ROOT, "gpurun_out", "parity_r06.json"))
    try:
        os.makedirs(os.path.dirname(out), exist_ok=True)
        with open(out, "w") as f:
            json.dump({"tolerances": {"value_rel": helpers.VALUE_RTOL, "action_abs": helpers.ACT_ATOL},
                       "exit_status": int(exitstatus), "cases": helpers.PARITY}, f, indent=1, sort_keys=True)
    except OSError:
        pass


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The parity table goes to STDOUT as well, so that the log of whoever runs `pytest -m gpu` carries the numbers (the
    JSON above is a file on the GPU box)."""
    try:
        from tests import helpers
    except Exception:
        return
    if not helpers.PARITY:
        return
    tr = terminalreporter
    tr.write_sep("=", "parity report: worst |HIP - reference/oracle| per case (gates: value_rel, action_abs, prev_mean_abs < 1e-4)")
    cols = ("value_rel", "mean_abs", "std_abs", "action_abs", "prev_mean_abs", "elite_swaps", "plans")
    tr.write_line(f"{'case':58s} " + " ".join(f"{c:>13s}" for c in cols))
    for key in sorted(helpers.PARITY):
        m = helpers.PARITY[key]
        cells = []
        for c in cols:
            v = m.get(c)
            cells.append(f"{'':>13s}" if v is None else (f"{v:13d}" if isinstance(v, int) else f"{v:13.2e}"))
        tr.write_line(f"{key[:58]:58s} " + " ".join(cells))
