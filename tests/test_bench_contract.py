"""bench.py's job contract with 2 REAL ranks on CPU (gloo) and a planner stand-in (TDMPC2_BENCH_STUB=1, tests/bench_stub.py):
the launch line the driver uses for N > 1, rank 0 alone prints ONE JSON line, `n_gpus` / `parallelism` / `value` follow
the world size, every rank takes part in the c5 leg and its all_reduce(MAX), the headline region times EXACTLY --steps and a
second region of >= 2 s is reported beside it (extra.long_region).
No kernel runs here: the numbers are the stand-in's; the logic around them is bench.py's own."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(nproc, args, extra_env=None, port=29641):
    env = dict(os.environ, TDMPC2_BENCH_STUB="1", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    return [ln for ln in p.stdout.splitlines() if ln.strip().startswith("{")], p.stderr


def test_two_ranks_print_one_line_with_whole_job_throughput_and_run_the_c5_leg():
    lines, err = _run(2, ["--gpus", "2", "--steps", "5", "--warmup", "2", "--envs", "16"], {"TDMPC2_BENCH_EXACT_STEPS": "1"})
    assert len(lines) == 1, (lines, err[-1500:])  # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["warmup"] == 2
    assert d["scaling"] == "weak" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["unit"] == "plans/s" and d["metric"].startswith("plan() calls/sec")
    assert d["config"]["parallelism"] == "env-sharded x2" and d["config"]["envs_per_gpu"] == 16
    # whole-job aggregate: plans of BOTH ranks over the max-over-ranks time
    assert d["value"] == pytest.approx(2 * 16 * 5 / (d["ms_per_step"] * 5 / 1e3), rel=1e-3)
    assert d["ms_per_step"] >= 4.0  # the stand-in sleeps 4 ms per step
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    c5 = d["extra"]["configs"]["c5"]
    assert "error" not in c5, c5
    assert c5["n_gpus"] == 2 and c5["config"]["envs_per_gpu"] == 64 and "128 envs" in c5["config"]["workload"]
    assert c5["value"] == pytest.approx(2 * 64 * c5["steps"] / (c5["ms_per_step"] * c5["steps"] / 1e3), rel=1e-2)
    # diagnosable first N > 1 record: per-rank step times and the world size the process group saw
    assert d["extra"]["world_size_seen"] == {"env": 2, "process_group": 2, "backend": "gloo"}
    r = d["extra"]["ms_per_step_over_ranks"]
    assert 4.0 <= r["min"] <= r["max"] == pytest.approx(d["ms_per_step"], rel=1e-3)


def test_headline_times_exactly_the_requested_steps_and_a_long_region_follows():
    """The driver's consistency check compares `steps` with its own --steps: the headline region times exactly that many; the
    >= 2 s region (an outside GPU-activity sampler needs it) is a second measurement, reported under extra.long_region."""
    lines, _ = _run(1, ["--gpus", "1", "--steps", "10", "--warmup", "2", "--envs", "8", "--skip-extra-configs"], port=29643)
    d = json.loads(lines[0])
    assert d["steps"] == 10 and "steps_requested" not in d
    lr = d["extra"]["long_region"]
    assert lr["steps"] > 10 and lr["steps"] * lr["ms_per_step"] / 1e3 >= 1.8 and lr["seconds"] >= 1.8
    assert lr["value"] == pytest.approx(8 * lr["steps"] / lr["seconds"], rel=1e-2)
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "env-sharded x1"


def test_world_size_must_match_gpus():
    env = dict(os.environ, TDMPC2_BENCH_STUB="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=120)
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)
