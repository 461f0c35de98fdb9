"""Shared comparison helpers for the parity tests."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# Tolerances (fp32 path; north_star: "within 1e-4 fp32").  Trajectory values go
# through symexp, so they are compared relative to max(1, |v|).
VALUE_RTOL = 1e-4
ACT_ATOL = 1e-4


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, f"{name}.npz")))


def value_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))))


def elite_sets_equal(idx_a, idx_b):
    return set(np.asarray(idx_a).tolist()) == set(np.asarray(idx_b).tolist())


def boundary_gap(value, k):
    """Relative gap between the k-th and (k+1)-th largest value: if tiny, an
    elite swap between two fp32 implementations is legitimate."""
    v = np.sort(np.asarray(value, np.float64))[::-1]
    return float((v[k - 1] - v[k]) / max(1.0, abs(v[k - 1])))


# ---- parity report: every -m gpu comparison records its worst errors here; tests/conftest.py writes the collection to
# gpurun_out/parity_<round>.json at the end of the session (committed under profiles/ as the round's parity evidence)
PARITY = {}


def record_parity(key, **metrics):
    """Keep the worst (largest) value seen for every metric of `key`."""
    cur = PARITY.setdefault(key, {})
    for k, v in metrics.items():
        v = float(v) if not isinstance(v, (str, bool, int)) else v
        if isinstance(v, float):
            cur[k] = max(cur.get(k, 0.0), v)
        elif isinstance(v, int) and not isinstance(v, bool):
            cur[k] = cur.get(k, 0) + v
        else:
            cur[k] = v
