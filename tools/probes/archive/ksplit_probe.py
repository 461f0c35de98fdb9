"""g_gemm_w's K-split tail (TDMPC2_TUNE_KSPLIT = 1) on one configuration: repeated estimate_value calls against the whole-tile
result, with the handle's fault bookkeeping after every call.  usage: python tools/probes/ksplit_probe.py <case> <E> [calls]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests.helpers import value_err  # noqa: E402
from tests.test_gpu_layered import _ksplit_inputs  # noqa: E402

name, E = sys.argv[1], int(sys.argv[2])
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 4
planner, args, kw = _ksplit_inputs(name, E)
planner.set_ksplit(0)
whole = planner.estimate_value(*args, **kw).clone()
whole2 = planner.estimate_value(*args, **kw).clone()
torch.cuda.synchronize()
print(f"{name} E={E}: whole tiles repeatable: {torch.equal(whole, whole2)}; faults {planner.fault_info()}")
planner.set_ksplit(1)
prev = None
for i in range(calls):
    v = planner.estimate_value(*args, **kw).clone()
    torch.cuda.synchronize()
    fi = planner.fault_info()
    err = value_err(v.cpu().numpy(), whole.cpu().numpy())
    same = None if prev is None else bool(torch.equal(v, prev))
    bad_rows = (v - whole).abs().gt(1e-3).nonzero()
    print(f"  split call {i}: rel err vs whole {err:.3e}  same as previous call: {same}  faults_total {fi['faults_total']} degraded {fi['degraded']}"
          f"  rows off by > 1e-3: {bad_rows.shape[0]}" + (f" (first {bad_rows[0].tolist()}, last {bad_rows[-1].tolist()})" if bad_rows.shape[0] else ""))
    prev = v
planner.close()
