"""CPU: the bench line contract (one JSON object with the driver's keys plus `roofline` and `cpu_baseline`), checked on the
line committed from the last MI355X run, and that bench.py refuses to run without a GPU instead of falling back."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _latest_bench_line():
    prof = os.path.join(ROOT, "profiles")
    names = sorted(n for n in os.listdir(prof) if n.endswith("_bench.json"))
    assert names, "no committed bench line under profiles/"
    with open(os.path.join(prof, names[-1])) as f:
        return names[-1], json.loads(f.read())


def test_committed_bench_line_has_the_contract_keys():
    name, d = _latest_bench_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, (name, k)
    assert d["unit"] == "plans/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # value and per-step time agree: plans/s = envs * steps / elapsed
    envs = d["config"]["envs_per_gpu"] * d["n_gpus"]
    assert abs(d["value"] - envs / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-3
    # the as-written FLOP figure is the one DESIGN.md section 1 states
    sys.path.insert(0, ROOT)
    import bench
    from tdmpc2_amd.config import named_config

    cfg = named_config("c2")
    assert abs(bench.flops_plan(cfg, 6) / 1e9 - 47.74) < 0.01
    assert abs(bench.flops_rollout_launch(cfg, 1) / 1e9 - 7.93) < 0.01
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1


def test_bench_refuses_to_run_without_a_gpu():
    import torch

    if torch.cuda.is_available():
        return
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True,
                       text=True, timeout=300)
    assert p.returncode != 0 and "no CPU path" in (p.stderr + p.stdout)
