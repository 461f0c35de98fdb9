"""plan() calls/sec on MI355X — BASELINE.json's metric.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E] [--config c2] [--iterations 6]

A "step" is one pass of the hot path over one batch of synthetic input: E independent plans
(synthetic SimNorm latents, random-init weights of the named architecture, in-kernel Philox noise —
everything of TDMPC2._plan after encode()).  Inputs are resident in HBM before the timed region.
For N > 1 launch with torch.distributed.run (one rank per GPU, RCCL); environments are sharded,
E per GPU (weak scaling), no collective in the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from tdmpc2_amd import synth  # noqa: E402
from tdmpc2_amd.config import get_discount, named_config  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
MIN_TIMED_S = 2.0           # the timed region is extended to at least this (see main)
F16_MFMA_PEAK_TFLOPS = 2500.0  # same guide: BF16/F16 MFMA, dense (the 2:1-sparsity figure is NOT used)


def per_row_macs(cfg):
    L, M, A, T, B = cfg.latent_dim, cfg.mlp_dim, cfg.action_dim, cfg.task_dim, cfg.num_bins
    dyn = (L + A + T) * M + M * M + M * L
    rew = (L + A + T) * M + M * M + M * B
    pi = (L + T) * M + M * M + 2 * A * M
    return dyn, rew, pi


def flops_rollout_launch(cfg, n_envs):
    """As-written FLOPs of one CEM iteration's _estimate_value for n_envs plans:
    all num_q Q heads counted, elementwise work excluded."""
    dyn, rew, pi = per_row_macs(cfg)
    return n_envs * 2 * cfg.num_samples * (cfg.horizon * (rew + dyn) + pi + cfg.num_q * rew)


def flops_rollout_executed(cfg, n_envs, fused):
    """FLOPs the kernels actually issue for the same launch: 2 of the num_q Q heads (the two the reference's
    `Q(..., return_type='avg')` keeps, world_model.py:175-178, are drawn before the launch), and in the fused family
    the z0 half of both first layers at t = 0 is computed once per environment instead of once per sample."""
    dyn, rew, pi = per_row_macs(cfg)
    per_row = cfg.horizon * (rew + dyn) + pi + 2 * rew
    f = n_envs * 2 * cfg.num_samples * per_row
    if fused:
        f -= n_envs * 2 * (cfg.num_samples - 1) * 2 * (cfg.latent_dim + cfg.task_dim) * cfg.mlp_dim
    return f


def flops_plan(cfg, iterations):
    dyn, rew, pi = per_row_macs(cfg)
    pitraj = cfg.num_pi_trajs * (cfg.horizon * pi + (cfg.horizon - 1) * dyn)
    return 2 * pitraj + iterations * flops_rollout_launch(cfg, 1)


# TDMPC2_BENCH_STUB=1: CPU dry run of the JOB logic (argument contract, rank-0-only line, max-over-ranks timing, the c5 leg's
# collective) with 2 real ranks over gloo and a planner stand-in (tests/bench_stub.py); never a measurement
# (tests/test_bench_contract.py).  Everything that needs kernels is skipped in that mode.
STUB = os.environ.get("TDMPC2_BENCH_STUB") == "1"


def _planner_cls():
    if STUB:
        from tests.bench_stub import StubPlanner
        return StubPlanner
    from tdmpc2_amd.native import NativePlanner
    return NativePlanner


def _sync(device):
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def box_under_load(queue_work, device, with_props=False):
    """What box is this, per LEG (clocks were sampled once per run, so a slow box could not be told from slow code
    inside the record)?  The kernels are power-managed: boxes of the same pool differ by up to 15 % in every figure.  Outside the
    timed region: `queue_work()` enqueues about half a second of the leg's own steps, the clocks / socket power the driver
    reports are read while they run."""
    import subprocess
    try:
        queue_work()
        smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showmaxpower"], capture_output=True, text=True, timeout=20)
        _sync(device)
        keep = {}
        for line in smi.stdout.splitlines():
            low = line.lower()
            for key, pat in (("sclk", "sclk"), ("mclk", "mclk"), ("fclk", "fclk"), ("power_w", "current socket"),
                             ("power_cap_w", "max graphics package power")):
                if pat in low and ":" in line and key not in keep:  # "GPU[0]  : sclk clock level: S: (2100Mhz)"
                    keep[key] = line.split(":", 1)[1].strip()[:90]
        if with_props:
            pr = torch.cuda.get_device_properties(device)
            for k in ("name", "gcnArchName", "multi_processor_count", "clock_rate", "memory_clock_rate", "L2_cache_size"):
                v = getattr(pr, k, None)
                if v is not None:
                    keep["prop_" + k] = v if isinstance(v, (int, float)) else str(v)[:60]
        return keep or {"raw": smi.stdout[:300]}
    except Exception as ex:  # no rocm-smi, no permission: the line is still valid
        return {"error": repr(ex)[:120]}


def disc_pow_rows(cfg, n_envs, device):
    g = get_discount(cfg, cfg.episode_length)
    d, vals = 1, []
    for _ in range(cfg.horizon + 1):
        vals.append(float(d))
        d = d * g
    return torch.tensor(vals, dtype=torch.float32, device=device).repeat(n_envs, 1).contiguous()


def _planner_for(name, E, I, device):
    """(cfg, planner with synthetic weights bound, plan inputs) of workload `name` -- what the timed legs build."""
    from tdmpc2_amd.native import NativePlanner

    cfg = named_config(name)
    sd_np = synth.make_state_dict(cfg, seed=0)
    sd = {k: torch.as_tensor(v).to(device) for k, v in sd_np.items() if not k.startswith("_encoder.")}
    planner = NativePlanner(cfg, I, device, max_envs=E)
    planner.bind_state_dict(sd)
    z0 = torch.as_tensor(synth.make_latents(cfg, E, seed=1000)).to(device)
    emb = mask = None
    if cfg.multitask:
        tasks = torch.arange(E) % len(cfg.tasks)
        w = torch.as_tensor(sd_np["_task_emb.weight"])[tasks]
        n = w.norm(dim=1, keepdim=True)
        emb = torch.where(n > 1.0, w / (n + 1e-7), w).to(device).contiguous()
        mask = torch.as_tensor(sd_np["_action_masks"])[tasks].to(device).contiguous()
    return cfg, planner, dict(z0=z0, disc=disc_pow_rows(cfg, E, device), emb=emb, mask=mask)


def traffic_child(name, E, I, steps):
    """Child mode (run under `rocprofv3 --pmc` by measured_traffic): the same workload, a few untimed steps, no output.
    Fused family: whole plans (the parent reads the ks_rollout launches of the throughput grid).  Layered family: the
    rollout STAGE alone (tdmpc2_plan_estimate_value = the GEMM / row-kernel sequence of one CEM iteration's
    _estimate_value, which is what `roofline` times), so that every g_gemm* / l_* dispatch of the trace belongs to a stage."""
    device = torch.device("cuda", 0)
    cfg, planner, x = _planner_for(name, E, I, device)
    prev = torch.zeros(E, cfg.horizon, cfg.action_dim, device=device)
    warm = torch.zeros(E, dtype=torch.uint8, device=device)
    if planner.path == 1:
        for i in range(steps + 1):
            planner.plan(x["z0"], x["disc"], prev, warm, task_emb=x["emb"], act_mask=x["mask"], seed=i)
    else:
        acts = torch.rand(E, cfg.horizon, cfg.num_samples, cfg.action_dim, device=device) * 2 - 1
        eps = torch.randn(E, cfg.num_samples, cfg.action_dim, device=device)
        qidx = torch.tensor([[0, 1]] * E, dtype=torch.int32, device=device)
        for i in range(steps):
            planner.estimate_value(x["z0"], x["disc"], acts, eps, qidx, task_emb=x["emb"], act_mask=x["mask"])
    _sync(device)
    planner.close()


def measured_traffic(name, E, I, fused, workgroups, steps=2, timeout=420):
    """HBM-side bytes per rollout launch / stage of THIS workload on THIS box: two `rocprofv3 --kernel-trace --pmc` children of
    this script (FETCH_SIZE and WRITE_SIZE need separate passes: TCC has 4 slots, MI355X_MICROARCH.md), run after the timed
    region.  FETCH_SIZE is in KiB and under-reports wide coalesced reads by 2x on gfx950 (same guide, HBM section): x 1024 x 2;
    WRITE_SIZE KiB x 1024 (uncalibrated).  Returns (bytes per launch or None, detail dict)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if shutil.which("rocprofv3") is None:
        return None, {"error": "rocprofv3 not on PATH"}
    tmp = tempfile.mkdtemp(prefix="tdmpc2_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    per = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "t", "--",
                   sys.executable, os.path.abspath(__file__), "--traffic-child", name, "--envs", str(E), "--iterations", str(I),
                   "--steps", str(steps)]
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, {"error": f"rocprofv3 --pmc {counter} rc={p.returncode}: {(p.stderr or p.stdout)[-200:]}"}
            total, launches = 0.0, set()
            with open(files[0]) as f:
                for r in csv.DictReader(f):
                    if r["Counter_Name"] != counter:
                        continue
                    kn = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").strip()
                    wgs = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
                    if fused:
                        if not (kn.startswith("ks_rollout<") and wgs == workgroups):
                            continue
                        launches.add(r["Dispatch_Id"])
                    else:
                        if not (kn.startswith("g_gemm") or kn.startswith("l_")):
                            continue
                    total += float(r["Counter_Value"])
            n = len(launches) if fused else steps
            if n == 0:
                return None, {"error": f"no matching dispatches in the {counter} pass"}
            per[counter] = total / n * 1024.0
        fetch, write = 2.0 * per["FETCH_SIZE"], per["WRITE_SIZE"]
        return fetch + write, {"fetch_bytes": round(fetch), "write_bytes": round(write), "launches_per_pass": n,
                               "source": "this run: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE children of bench.py "
                                         "on the same box after the timed region"}
    except Exception as ex:
        return None, {"error": repr(ex)[:200]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def action_mse_vs_reference(device, path, prec):
    """The second half of BASELINE.json's metric ("action MSE vs ref"): the planner replays the recorded noise tape of
    the committed golden case for the benched model (c2: dog-run 5M, 2 envs, the reference's 8 iterations for A >= 20)
    and its actions are compared with the actions the reference's OWN planner code produced on the same inputs
    (tests/golden/c2.npz, made by oracle/make_golden.py).  Checker use of the test infrastructure, outside the timed
    region; nothing here is measured for speed."""
    import numpy as np
    from oracle import cases
    from tdmpc2_amd.native import NativePlanner

    c = cases.build_case("c2")
    g = np.load(os.path.join(ROOT, "tests", "golden", "c2.npz"))
    cfg, E = c["cfg"], c["n_envs"]
    pl = NativePlanner(cfg, c["iterations"], device, max_envs=E, path=path, precision=prec)
    pl.bind_state_dict({k: torch.as_tensor(v) for k, v in c["sd"].items()})
    tape = {k: torch.as_tensor(v).to(device).contiguous() for k, v in c["tape"].items()}
    z0 = torch.as_tensor(c["z0"]).to(device)
    prev = torch.as_tensor(c["prev_mean"]).to(device).clone()
    t0 = torch.as_tensor(c["t0"].astype(np.uint8)).to(device)
    a = pl.plan(z0, disc_pow_rows(cfg, E, device), prev, t0, eval_mode=False, tape=tape).cpu().numpy()
    pl.close()
    d = a.astype(np.float64) - g["action"].astype(np.float64)
    return {"action_mse_vs_reference": float((d ** 2).mean()), "action_max_abs_diff": float(np.abs(d).max()),
            "case": "tests/golden/c2.npz: the reference's own planner code on the same weights, latents and noise tape "
                    f"({E} envs x {c['iterations']} CEM iterations)"}


def torch_eager_gpu_baseline(cfg, iterations, sd_np, device, n_plans=5):
    """The same restatement of the reference's planner math, run as stock PyTorch-ROCm eager ops on the SAME MI355X
    (what `python evaluate.py compile=false` does with the reference): a reported reference point, like cpu_baseline."""
    from oracle import planner_oracle as po

    model = po.OracleModel(cfg, {k: torch.as_tensor(v) for k, v in sd_np.items()}, device=device)
    z0 = torch.as_tensor(synth.make_latents(cfg, 1, seed=1)).to(device)
    tape = {k: v.to(device) for k, v in po.env_tape(synth.make_noise_tape(cfg, 1, iterations, seed=2), 0).items()}
    disc = get_discount(cfg, cfg.episode_length)
    prev = torch.zeros(cfg.horizon, cfg.action_dim, device=device)
    task = 0 if cfg.multitask else None
    disc = torch.tensor(disc, device=device) if cfg.multitask else disc
    with torch.no_grad():
        for _ in range(2):
            a, prev, _ = po.plan(model, z0=z0, tape=tape, prev_mean=prev, t0=False, eval_mode=False, task=task, discount=disc,
                                 iterations=iterations)
        _sync(device)
        t0 = time.perf_counter()
        for _ in range(n_plans):
            a, prev, _ = po.plan(model, z0=z0, tape=tape, prev_mean=prev, t0=False, eval_mode=False, task=task, discount=disc,
                                 iterations=iterations)
        _sync(device)
        el = time.perf_counter() - t0
    return {"value": round(n_plans / el, 2), "unit": "plans/s", "ms_per_plan": round(1e3 * el / n_plans, 2),
            "what": f"oracle restatement of the reference planner as PyTorch-ROCm {torch.__version__} eager ops on this GPU, "
                    f"1 env, {n_plans} sequential plans"}


def torch_compile_child(name, iterations, n_plans=20):
    """Child mode: the reference's DEFAULT execution mode (config.yaml:74 `compile: true`; tdmpc2.py:41-55:
    torch.compile(self._plan, mode="reduce-overhead")) applied to the oracle restatement of `_plan` on this GPU; prints one
    JSON object.  Reported baseline only."""
    from oracle import planner_oracle as po

    device = torch.device("cuda", 0)
    cfg = named_config(name)
    sd_np = synth.make_state_dict(cfg, seed=0)
    model = po.OracleModel(cfg, {k: torch.as_tensor(v) for k, v in sd_np.items()}, device=device)
    z0 = torch.as_tensor(synth.make_latents(cfg, 1, seed=1)).to(device)
    tape = {k: v.to(device) for k, v in po.env_tape(synth.make_noise_tape(cfg, 1, iterations, seed=2), 0).items()}
    disc = get_discount(cfg, cfg.episode_length)

    def plan_fn(z, prev):
        a, pm, _ = po.plan(model, z0=z, tape=tape, prev_mean=prev, t0=False, eval_mode=False, task=None, discount=disc,
                           iterations=iterations)
        return a, pm

    t_c = time.perf_counter()
    fn = torch.compile(plan_fn, mode="reduce-overhead")
    prev = torch.zeros(cfg.horizon, cfg.action_dim, device=device)
    with torch.no_grad():
        for _ in range(3):  # compile + CUDA-graph capture
            torch.compiler.cudagraph_mark_step_begin()
            a, pm = fn(z0, prev)
            prev = pm.clone()
        _sync(device)
        compile_s = time.perf_counter() - t_c
        t0 = time.perf_counter()
        for _ in range(n_plans):
            torch.compiler.cudagraph_mark_step_begin()
            a, pm = fn(z0, prev)
            prev = pm.clone()
        _sync(device)
        el = time.perf_counter() - t0
    print(json.dumps({"value": round(n_plans / el, 2), "unit": "plans/s", "ms_per_plan": round(1e3 * el / n_plans, 3),
                      "compile_s": round(compile_s, 1), "finite": bool(torch.isfinite(a).all()),
                      "what": f"torch.compile(mode='reduce-overhead') of the oracle restatement of the reference planner "
                              f"(torch {torch.__version__}, Inductor / Triton-ROCm, graph replay), 1 env, {n_plans} sequential plans"}))


def torch_compile_baseline(name, iterations, timeout=240):
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--compile-child", name, "--iterations", str(iterations)],
                           env=env, capture_output=True, text=True, timeout=timeout)
        for line in reversed(p.stdout.strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"value": None, "status": "unavailable", "error": (p.stderr or p.stdout)[-300:]}
    except subprocess.TimeoutExpired:
        return {"value": None, "status": f"no result within {timeout} s (compilation of the 6-iteration plan did not finish)"}
    except Exception as ex:
        return {"value": None, "status": "unavailable", "error": repr(ex)[:200]}


def cpu_baseline(cfg, iterations, sd_np, budget_s=12.0):
    """The oracle (= the reference's planner math as plain torch CPU ops) timed on this box's host cores
    on a bounded sample of the same workload.  Test infrastructure used as a reported baseline only."""
    from oracle import planner_oracle as po

    try:
        threads = len(os.sched_getaffinity(0))
    except AttributeError:
        threads = os.cpu_count() or 1
    threads = max(1, min(threads, 64))  # torch's intra-op pool does not scale past the physical cores
    torch.set_num_threads(threads)
    model = po.OracleModel(cfg, {k: torch.as_tensor(v) for k, v in sd_np.items()})
    z0 = synth.make_latents(cfg, 1, seed=1)
    tape = po.env_tape(synth.make_noise_tape(cfg, 1, iterations, seed=2), 0)
    disc = get_discount(cfg, cfg.episode_length)
    prev = torch.zeros(cfg.horizon, cfg.action_dim)

    def one(t0):
        nonlocal prev
        a, prev, _ = po.plan(model, z0=torch.as_tensor(z0), tape=tape, prev_mean=prev, t0=t0, eval_mode=False,
                             task=0 if cfg.multitask else None,
                             discount=torch.tensor(disc) if cfg.multitask else disc, iterations=iterations)

    with torch.no_grad():
        tw = time.perf_counter()
        one(True)  # warm-up
        print(f"[bench] cpu baseline: {threads} threads, warm-up plan {time.perf_counter() - tw:.2f} s", file=sys.stderr,
              flush=True)
        n, t_start = 0, time.perf_counter()
        while True:
            one(False)
            n += 1
            el = time.perf_counter() - t_start
            if (el >= budget_s and n >= (3 if el < 4 * budget_s else 1)) or n >= 64:
                break
    # time per plan of this restatement / of the reference's own TDMPC2._plan run verbatim (oracle/ref_runner.py): MEASURED here
    # when the reference tree is on this machine (the build container).  The reference is Python and may not travel to the GPU box
    # in any form, so there the ratio comes from the committed record of the one place where both exist
    # (profiles/port_vs_reference.json, made by oracle/time_port_vs_reference.py: same inputs, same thread count, interleaved)
    pvr = {"value": 1.0, "source": "constant (no record for this configuration)"}
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "port_vs_reference.json")))
        key = "c1" if cfg.action_dim < 20 and not cfg.multitask else "c2"
        if not cfg.multitask and cfg.latent_dim == 512 and key in rec["cases"]:
            r = rec["cases"][key]
            pvr = {"value": r["port_vs_reference"],
                   "source": f"profiles/port_vs_reference.json ({key}: port {r['port_ms_per_plan']} ms, the reference's own _plan {r['reference_ms_per_plan']} ms "
                             f"per plan on {rec['threads']} threads of the build container) -- NOT measured on this machine: /root/reference is not here"}
    except Exception:
        pass
    if os.path.isdir("/root/reference/tdmpc2"):
        try:
            from oracle import ref_runner

            sd_t = {k: torch.as_tensor(v) for k, v in sd_np.items()}
            ref_ms, m = ref_runner.time_reference_plan(
                cfg, sd_t, z0=z0[0], tape=tape, task=0 if cfg.multitask else None, iterations=iterations,
                discount=torch.tensor([disc] * len(cfg.tasks)) if cfg.multitask else disc, budget_s=min(budget_s, 6.0))
            pvr = {"value": round((1e3 * el / n) / ref_ms, 3), "source": f"measured in this run: {m} plans of the reference's own _plan "
                                                                        f"(oracle/ref_runner.time_reference_plan), {ref_ms:.1f} ms each"}
        except Exception as ex:
            pvr["error"] = repr(ex)[:160]
    return {"value": round(n / el, 3), "unit": "plans/s", "cores": int(torch.get_num_threads()), "kind": "port",
            "reference_estimate": round(n / el * pvr["value"], 3),  # what the reference's own _plan would do here, by that ratio
            "port_vs_reference": pvr["value"], "port_vs_reference_source": pvr["source"], **({"port_vs_reference_error": pvr["error"]} if "error" in pvr else {}),
            "sample": f"{n} sequential plan() calls of the same workload (1 env, recorded noise tape) after 1 warm-up, "
                      f"{el:.1f} s wall, torch {torch.__version__} CPU fp32",
            "ms_per_plan": round(1e3 * el / n, 2)}


def c5_leg(device, rank, world, fence, steps=2, envs_per_gpu=64):
    """BASELINE.json configs[4]: mt80 317M (c4's model, H5 N1024), 64 vectorised synthetic envs per GPU, env-sharded over the
    ranks of this job (512 envs at 8 GPUs) -- run by EVERY rank after the main measurement when the job has more than one
    GPU; same timing protocol (barrier + synchronize on both sides, max over ranks)."""
    import torch.distributed as dist
    NativePlanner = _planner_cls()

    cfg = named_config("c4")
    I = cfg.iterations + 2 * int(cfg.action_dim >= 20)
    E = envs_per_gpu
    tasks = (torch.arange(E) + rank * E) % len(cfg.tasks)
    planner = NativePlanner(cfg, I, device, max_envs=E)
    if STUB:  # (no 317M of synthetic weights for a dry run of the job logic)
        emb = torch.zeros(E, cfg.task_dim)
        mask = torch.ones(E, cfg.action_dim)
    else:
        sd_np = synth.make_state_dict(cfg, seed=0)  # identical on every rank (seeded): no broadcast needed for the leg
        sd = {k: torch.as_tensor(v).to(device) for k, v in sd_np.items() if not k.startswith("_encoder.")}
        planner.bind_state_dict(sd)
        w = torch.as_tensor(sd_np["_task_emb.weight"])[tasks]
        n = w.norm(dim=1, keepdim=True)
        emb = torch.where(n > 1.0, w / (n + 1e-7), w).to(device).contiguous()
        mask = torch.as_tensor(sd_np["_action_masks"])[tasks].to(device).contiguous()
        del sd_np
    z0 = torch.as_tensor(synth.make_latents(cfg, E, seed=3000 + rank)).to(device)
    disc = disc_pow_rows(cfg, E, device)
    prev = torch.zeros(E, cfg.horizon, cfg.action_dim, device=device)
    warm = torch.zeros(E, dtype=torch.uint8, device=device)
    out = torch.empty(E, cfg.action_dim, device=device)
    planner.plan(z0, disc, prev, torch.ones_like(warm), task_emb=emb, act_mask=mask, seed=(rank << 32) + 1, out=out)
    planner.set_profiling(steps * I)
    fence()
    t1 = time.perf_counter()
    for i in range(steps):
        planner.plan(z0, disc, prev, warm, task_emb=emb, act_mask=mask, seed=(rank << 32) + 10 + i, out=out)
    fence()
    el = time.perf_counter() - t1
    ms, nl = planner.profile_read()
    planner.close()
    t = torch.tensor([el], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el = float(t.item())
    launch_s = (ms / 1e3) / max(nl, 1)
    ach = flops_rollout_launch(cfg, E) / launch_s / 1e12
    return {"value": round(world * E * steps / el, 2), "unit": "plans/s (whole job)", "n_gpus": world, "steps": steps,
            "ms_per_step": round(1e3 * el / steps, 2), "scaling": "weak",
            "config": {"workload": f"c5: mt80 317M (L{cfg.latent_dim} M{cfg.mlp_dim} nq{cfg.num_q}), plan() H={cfg.horizon} "
                                   f"N={cfg.num_samples} I={I}, {E} envs per GPU x {world} GPUs = {world * E} envs, env-sharded, "
                                   "no collective in the timed region", "envs_per_gpu": E},
            "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": F16_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s (this rank)",
                         "frac": round(ach / F16_MFMA_PEAK_TFLOPS, 4), "avg_stage_ms": round(1e3 * launch_s, 3), "traffic": None}}


def config_leg(name, E, steps, device, rank=0, single_env=True):
    """A short measurement of another BASELINE.json configuration (c3: mt30 48M, one plan per task id; c4: mt80 317M, H5
    N1024) in the same process, reported under extra.configs with its own roofline -- so that the driver's default run
    carries numbers for configs[2] and configs[3] too (missing #7).  Same rules as the main line: inputs
    resident in HBM, random-init weights of the named architecture, in-kernel Philox noise, HIP-event timing of the
    rollout stage on the launch stream, nothing skipped."""
    from tdmpc2_amd.native import NativePlanner

    cfg = named_config(name)
    I = cfg.iterations + 2 * int(cfg.action_dim >= 20)
    t_leg = time.perf_counter()
    sd_np = synth.make_state_dict(cfg, seed=0)
    sd = {k: torch.as_tensor(v).to(device) for k, v in sd_np.items() if not k.startswith("_encoder.")}
    planner = NativePlanner(cfg, I, device, max_envs=E)
    planner.bind_state_dict(sd)
    family = {1: "fused", 2: "layered"}[planner.path]
    split = planner.precision == 2
    z0 = torch.as_tensor(synth.make_latents(cfg, E, seed=2000 + rank)).to(device)
    disc = disc_pow_rows(cfg, E, device)
    emb = mask = None
    if cfg.multitask:
        tasks = torch.arange(E) % len(cfg.tasks)
        w = torch.as_tensor(sd_np["_task_emb.weight"])[tasks]
        n = w.norm(dim=1, keepdim=True)
        emb = torch.where(n > 1.0, w / (n + 1e-7), w).to(device).contiguous()
        mask = torch.as_tensor(sd_np["_action_masks"])[tasks].to(device).contiguous()
    del sd_np
    prev = torch.zeros(E, cfg.horizon, cfg.action_dim, device=device)
    warm = torch.zeros(E, dtype=torch.uint8, device=device)
    cold = torch.ones(E, dtype=torch.uint8, device=device)
    out = torch.empty(E, cfg.action_dim, device=device)
    planner.plan(z0, disc, prev, cold, task_emb=emb, act_mask=mask, seed=1, out=out)
    planner.plan(z0, disc, prev, warm, task_emb=emb, act_mask=mask, seed=2, out=out)
    planner.set_profiling(steps * I)
    _sync(device)
    t1 = time.perf_counter()
    for i in range(steps):
        planner.plan(z0, disc, prev, warm, task_emb=emb, act_mask=mask, seed=10 + i, out=out)
    _sync(device)
    el = time.perf_counter() - t1
    ms, n = planner.profile_read()
    finite = bool(torch.isfinite(out).all())
    leg_faults = planner.take_fault()
    dev_mib = planner.device_bytes / 2**20
    n_load = max(2, int(np.ceil(0.5 * steps / max(el, 1e-3))))  # about half a second of this leg's steps
    box = box_under_load(lambda: [planner.plan(z0, disc, prev, warm, task_emb=emb, act_mask=mask, seed=500 + i, out=out)
                                  for i in range(n_load)], device)
    planner.close()
    lat1 = lat1_tiles = None
    if single_env:  # the reference's own semantics: one environment, one plan at a time (evaluate.py:80)
        one = NativePlanner(cfg, I, device, max_envs=1)
        one.bind_state_dict(sd)
        z1, d1, p1 = z0[:1].contiguous(), disc[:1].contiguous(), torch.zeros(1, cfg.horizon, cfg.action_dim, device=device)
        e1 = emb[:1].contiguous() if emb is not None else None
        m1 = mask[:1].contiguous() if mask is not None else None
        o1 = torch.empty(1, cfg.action_dim, device=device)
        for i in range(2):
            one.plan(z1, d1, p1, warm[:1], seed=i, out=o1, task_emb=e1, act_mask=m1)
        _sync(device)
        t1 = time.perf_counter()
        for i in range(5):
            one.plan(z1, d1, p1, warm[:1], seed=10 + i, out=o1, task_emb=e1, act_mask=m1)
        _sync(device)
        lat1 = (time.perf_counter() - t1) / 5 * 1e3
        if family == "layered" and split:
            # the same single plan on the per-layer tiles of the batch path (TDMPC2_TUNE_FEWROW = 0): what the few-row path
            # (tdmpc2_amd/csrc/layered_mid.cuh, round 6) buys, in the same run on the same box
            one.set_fewrow(0)
            for i in range(2):
                one.plan(z1, d1, p1, warm[:1], seed=i, out=o1, task_emb=e1, act_mask=m1)
            _sync(device)
            t1 = time.perf_counter()
            for i in range(5):
                one.plan(z1, d1, p1, warm[:1], seed=10 + i, out=o1, task_emb=e1, act_mask=m1)
            _sync(device)
            lat1_tiles = (time.perf_counter() - t1) / 5 * 1e3
        one.close()
    launch_s = (ms / 1e3) / max(n, 1)
    peak = F16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
    ach = flops_rollout_launch(cfg, E) / launch_s / 1e12
    ach_x = flops_rollout_executed(cfg, E, family == "fused") / launch_s / 1e12
    return {
        "value": round(steps * E / el, 2), "unit": "plans/s", "steps": steps, "ms_per_step": round(1e3 * el / steps, 3), "finite": finite,
        "bounded_wait_faults": leg_faults, "box_under_load": box,
        **({"latency_ms_single_env": round(lat1, 3)} if lat1 is not None else {}),
        **({"latency_ms_single_env_per_layer_tiles": round(lat1_tiles, 3)} if lat1_tiles is not None else {}),
        "config": {"workload": f"{name}: {cfg.task} world model (L{cfg.latent_dim} M{cfg.mlp_dim} A{cfg.action_dim} nq{cfg.num_q} "
                               f"T{cfg.task_dim}), plan() H={cfg.horizon} N={cfg.num_samples} K={cfg.num_elites} P={cfg.num_pi_trajs} "
                               f"I={I}, {E} concurrent plans (one per task id, round-robin), random-init weights",
                   "envs": E, "iterations": I, "kernel_family": family, "device_MiB": round(dev_mib),
                   "gflop_per_plan_as_written": round(flops_plan(cfg, I) / 1e9, 1)},
        "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                     "achieved_executed": round(ach_x, 2), "traffic": None,
                     "kernel": ("g_gemm_s" if split else "g_gemm") + " + row kernels of one _estimate_value" if family == "layered" else "ks_rollout",
                     "launches_timed": n, "avg_stage_ms": round(1e3 * launch_s, 3)},
        "leg_wall_s": round(time.perf_counter() - t_leg, 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # 80 steps of 256 plans = a timed region of about two seconds (20 steps were half a second: too short for an outside
    # observer's GPU-activity sampler to see)
    ap.add_argument("--steps", type=int, default=80)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--envs", type=int, default=256, help="independent environments planned per GPU per step")
    ap.add_argument("--config", default="c2", help="c1 cheetah-run 5M | c2 dog-run 5M (BASELINE configs[1])")
    ap.add_argument("--iterations", type=int, default=6, help="CEM iterations (the metric is quoted at 6)")
    ap.add_argument("--path", default="auto", choices=["auto", "fused", "layered"],
                    help="kernel family (auto: fused for 512-wide models, layered otherwise)")
    ap.add_argument("--precision", default="auto", choices=["auto", "fp32", "split"],
                    help="contraction arithmetic: exact-fp32 MFMA or f16x2-split on the f16 matrix pipe (fp32-class "
                         "accuracy); auto = split")
    ap.add_argument("--rows-per-workgroup", type=int, default=0, help="fused split kernels: 0 auto, 32 or 64 sample rows")
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-extra-configs", action="store_true",
                    help="do not append the short c3 / c4 legs (extra.configs) to the default c2 line")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    ap.add_argument("--skip-traffic", action="store_true",
                    help="do not launch the rocprofv3 --pmc children that measure roofline.traffic (use when bench.py itself runs under a profiler)")
    ap.add_argument("--traffic-child", default=None, metavar="CONFIG", help=argparse.SUPPRESS)
    ap.add_argument("--compile-child", default=None, metavar="CONFIG", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.traffic_child:
        return traffic_child(args.traffic_child, args.envs, args.iterations, args.steps)
    if args.compile_child:
        return torch_compile_child(args.compile_child, args.iterations)

    t_boot = time.perf_counter()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus}")
    if STUB:
        device = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the planner has no CPU path")
        # TDMPC2_BENCH_ONE_GPU=1: every rank on cuda:0 -- two PROCESSES on the builder's single GPU, to run the N > 1 job logic over RCCL
        # itself (init, broadcast, all_reduce(MAX)) where no second GPU exists; a dry run of the collectives, never a measurement
        device = torch.device("cuda", 0 if os.environ.get("TDMPC2_BENCH_ONE_GPU") else local_rank)
        torch.cuda.set_device(device)
    import torch.distributed as dist
    # under torch.distributed.run (RANK set) the RCCL group is always initialised, also for one rank, so that the same
    # collective calls run at every N
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if STUB:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from tdmpc2_amd.dist import broadcast_state_dict
    NativePlanner = _planner_cls()

    cfg = named_config(args.config)
    I, E, K, W = args.iterations, args.envs, args.steps, args.warmup
    sd_np = synth.make_state_dict(cfg, seed=0)
    if os.environ.get("TDMPC2_BENCH_ZERO_WEIGHTS"):  # power probe only (profiles/README.md): all-zero operands, same instruction stream
        sd_np = {k: (np.zeros_like(v) if k.endswith(".weight") and v.ndim >= 2 else v) for k, v in sd_np.items()}
    # weights: rank 0's copy is broadcast over RCCL (one bucketed collective), outside the timed region
    sd = {k: (torch.as_tensor(v).to(device) if rank == 0 else torch.zeros(v.shape, dtype=torch.float32, device=device))
          for k, v in sd_np.items()}
    broadcast_state_dict(sd, src=0)
    path = {"auto": 0, "fused": 1, "layered": 2}[args.path]
    prec = {"auto": 0, "fp32": 1, "split": 2}[args.precision]
    planner = NativePlanner(cfg, I, device, max_envs=E, path=path, precision=prec)
    if os.environ.get("TDMPC2_FOLD_REFIT") and planner.path == 1:  # A/B knob: 0 never, 1 always, 2 auto (default)
        planner.set_fold_refit(int(os.environ["TDMPC2_FOLD_REFIT"]))
    family = {1: "fused", 2: "layered"}[planner.path]
    arith = {1: "fp32 MFMA (v_mfma_f32_32x32x2_f32)", 2: "f16x2 split (3x v_mfma_f32_32x32x16_f16, fp32 accumulate)"}[planner.precision]
    planner.bind_state_dict(sd)
    if args.rows_per_workgroup:
        planner.set_rows_per_workgroup(args.rows_per_workgroup)

    z0 = torch.as_tensor(synth.make_latents(cfg, E, seed=1000 + rank)).to(device)
    disc = disc_pow_rows(cfg, E, device)
    emb = mask = None
    if cfg.multitask:  # one plan per task id, round-robin (the reference plans the tasks one at a time)
        tasks = torch.arange(E) % len(cfg.tasks)
        w = torch.as_tensor(sd_np["_task_emb.weight"])[tasks]
        n = w.norm(dim=1, keepdim=True)
        emb = torch.where(n > 1.0, w / (n + 1e-7), w).to(device).contiguous()  # nn.Embedding(max_norm=1)
        mask = torch.as_tensor(sd_np["_action_masks"])[tasks].to(device).contiguous()
    prev = torch.zeros(E, cfg.horizon, cfg.action_dim, device=device)
    cold = torch.ones(E, dtype=torch.uint8, device=device)
    warm = torch.zeros(E, dtype=torch.uint8, device=device)
    out = torch.empty(E, cfg.action_dim, device=device)

    def step(i, flags):
        planner.plan(z0, disc, prev, flags, eval_mode=False, task_emb=emb, act_mask=mask, tape=None,
                     seed=(rank << 32) + i, out=out)

    def fence():
        _sync(device)
        if use_dist:
            dist.barrier()
            _sync(device)

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_boot:.1f}s] {msg}", file=sys.stderr, flush=True)

    log(f"planner ready: {planner.device_bytes / 2**20:.0f} MiB on device, E={E} I={I}")
    step(0, cold)
    _sync(device)
    log("first (cold) step done")
    for i in range(W):
        step(1 + i, warm)
    # The headline region times EXACTLY --steps steps (the driver is the authority on its own contract: its consistency check
    # compares `steps` with what it asked for).  Short requests -- the driver's --steps 20 is half a second -- are followed by a
    # second region of at least MIN_TIMED_S, reported as extra.long_region (value, steps, ms_per_step): long enough for an
    # outside GPU-activity sampler, and a check that the short figure is not a warm-clock artefact.
    planner.set_profiling(K * I)
    fence()
    t_start = time.perf_counter()
    for i in range(K):
        step(100 + i, warm)
    fence()
    elapsed = time.perf_counter() - t_start
    log(f"timed region: {K} steps in {elapsed:.3f} s")
    roll_ms, roll_n = planner.profile_read()
    planner.set_profiling(0)
    faults = planner.take_fault()  # bounded inter-workgroup waits that gave up in the timed region (0 on a healthy box)
    # the first N > 1 record should be diagnosable: per-rank step times (min / max over ranks) and the world size RCCL saw
    rank_ms = {"min": round(1e3 * elapsed / K, 3), "max": round(1e3 * elapsed / K, 3)}
    if use_dist:
        t = torch.tensor([elapsed, -elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rank_ms = {"min": round(-1e3 * float(t[1].item()) / K, 3), "max": round(1e3 * float(t[0].item()) / K, 3)}
        elapsed = float(t[0].item())
    assert faults > 0 or torch.isfinite(out).all()
    long_region = None
    if elapsed < MIN_TIMED_S and not os.environ.get("TDMPC2_BENCH_EXACT_STEPS"):
        K2 = int(np.ceil(1.1 * MIN_TIMED_S / max(elapsed / K, 1e-6)))
        if use_dist:
            tn = torch.tensor([K2], dtype=torch.int64, device=device)
            dist.all_reduce(tn, op=dist.ReduceOp.MAX)
            K2 = int(tn.item())
        K2 = min(K2, 100 * K)
        fence()
        t2 = time.perf_counter()
        for i in range(K2):
            step(10000 + i, warm)
        fence()
        el2 = time.perf_counter() - t2
        if use_dist:
            t = torch.tensor([el2], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
        faults += planner.take_fault()
        long_region = {"value": round(world * E * K2 / el2, 2), "unit": "plans/s", "steps": K2, "ms_per_step": round(1e3 * el2 / K2, 3),
                       "seconds": round(el2, 3), "note": f"a second timed region of >= {MIN_TIMED_S} s behind the headline's --steps"}
        log(f"long region: {K2} steps in {el2:.3f} s")

    # (digest of the last timed step's actions: same seed, same inputs -> A/B variants that claim identical sums can be compared)
    import hashlib
    extra = {"bounded_wait_faults": faults, "action_sha1": hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:16],
             "ms_per_step_over_ranks": rank_ms,
             "world_size_seen": {"env": world, "process_group": (dist.get_world_size() if use_dist else None),
                                 "backend": (dist.get_backend() if use_dist else None)}}
    if long_region is not None:
        extra["long_region"] = long_region
    if rank == 0 and world == 1 and not STUB:
        # per-leg clocks / socket power under the leg's own load (box_under_load)
        extra["box_under_load"] = box_under_load(lambda: [step(500 + i, warm) for i in range(15)], device, with_props=True)
    if rank == 0 and not STUB:
        # single-environment latency (the reference's E = 1 semantics), reported beside the throughput
        one = NativePlanner(cfg, I, device, max_envs=1, path=path, precision=prec)
        one.bind_state_dict(sd)
        z1, d1 = z0[:1].contiguous(), disc[:1].contiguous()
        e1 = emb[:1].contiguous() if emb is not None else None
        m1 = mask[:1].contiguous() if mask is not None else None
        p1 = torch.zeros(1, cfg.horizon, cfg.action_dim, device=device)
        o1 = torch.empty(1, cfg.action_dim, device=device)
        for i in range(2):
            one.plan(z1, d1, p1, warm[:1], seed=i, out=o1, task_emb=e1, act_mask=m1)
        _sync(device)
        t1 = time.perf_counter()
        for i in range(20):
            one.plan(z1, d1, p1, warm[:1], seed=10 + i, out=o1, task_emb=e1, act_mask=m1)
        _sync(device)
        extra["latency_ms_single_env"] = round((time.perf_counter() - t1) / 20 * 1e3, 3)
        if family == "fused" and one.precision == 2:
            # the same plan with one workgroup per 32-row tile (TDMPC2_TUNE_CLUSTER 0): what the cluster path buys
            one.set_cluster(0)
            for i in range(2):
                one.plan(z1, d1, p1, warm[:1], seed=i, out=o1, task_emb=e1, act_mask=m1)
            _sync(device)
            t1 = time.perf_counter()
            for i in range(5):
                one.plan(z1, d1, p1, warm[:1], seed=10 + i, out=o1, task_emb=e1, act_mask=m1)
            _sync(device)
            extra["latency_ms_single_env_no_cluster"] = round((time.perf_counter() - t1) / 5 * 1e3, 3)
            # ... and with ONE cluster of 8 workgroups per tile (TDMPC2_TUNE_CLUSTER 1, round 3's path): what the second cluster
            # per tile (reward chain beside the dynamics chain, cluster2_kernels.cuh) buys
            one.set_cluster(1)
            for i in range(2):
                one.plan(z1, d1, p1, warm[:1], seed=i, out=o1, task_emb=e1, act_mask=m1)
            _sync(device)
            t1 = time.perf_counter()
            for i in range(5):
                one.plan(z1, d1, p1, warm[:1], seed=10 + i, out=o1, task_emb=e1, act_mask=m1)
            _sync(device)
            extra["latency_ms_single_env_one_cluster_per_tile"] = round((time.perf_counter() - t1) / 5 * 1e3, 3)
            one.set_cluster(2)
        # the same from the observation on (WorldModel.encode in the library, tdmpc2_plan_run_obs): what one
        # TDMPC2.act() costs on the device
        try:
            one.bind_encoder({k: v for k, v in sd.items() if k.startswith("_encoder.state.")})
            ob = torch.as_tensor(synth.make_obs(cfg, 1, seed=3)).to(device)
            for i in range(2):
                one.plan_obs(ob, d1, p1, warm[:1], seed=i, out=o1, task_emb=e1, act_mask=m1)
            _sync(device)
            t1 = time.perf_counter()
            for i in range(5):
                one.plan_obs(ob, d1, p1, warm[:1], seed=20 + i, out=o1, task_emb=e1, act_mask=m1)
            _sync(device)
            extra["latency_ms_single_env_from_obs"] = round((time.perf_counter() - t1) / 5 * 1e3, 3)
            zb = torch.empty(1, cfg.latent_dim, device=device)
            _sync(device)
            t1 = time.perf_counter()
            for i in range(20):
                one.encode(ob, e1, out=zb)
            _sync(device)
            extra["encode_us_single_env"] = round((time.perf_counter() - t1) / 20 * 1e6, 1)
        except Exception as ex:
            extra["latency_ms_single_env_from_obs"] = {"error": repr(ex)}

        if family == "fused" and not cfg.multitask:
            # the training-side forward piece on the same kernels: TDMPC2._td_target at the reference's batch
            # (horizon 3 x batch_size 256 rows, config.yaml)
            try:
                R = cfg.horizon * 256
                nz = torch.as_tensor(synth.make_latents(cfg, R, seed=5)).to(device)
                rw, tm = torch.randn(R, device=device), torch.zeros(R, device=device)
                for i in range(3):
                    planner.td_target(nz, rw, tm, 0.99, seed=i)
                _sync(device)
                t1 = time.perf_counter()
                for i in range(20):
                    planner.td_target(nz, rw, tm, 0.99, seed=10 + i)
                _sync(device)
                extra["td_target_us_768_rows"] = round((time.perf_counter() - t1) / 20 * 1e6, 1)
            except Exception as ex:
                extra["td_target_us_768_rows"] = {"error": repr(ex)}
        if args.config == "c2":
            try:
                extra["parity"] = action_mse_vs_reference(device, path, prec)
            except Exception as ex:
                extra["parity"] = {"error": repr(ex)}
        if planner.precision == 2 and K >= 2:
            # companion measurement of the same workload with the exact-fp32 MFMA kernels (a few steps), so that one
            # bench line carries both arithmetic modes
            ex = NativePlanner(cfg, I, device, max_envs=E, path=path, precision=1)
            ex.bind_state_dict(sd)
            pe = torch.zeros_like(prev)
            for i in range(2):
                ex.plan(z0, disc, pe, warm, task_emb=emb, act_mask=mask, seed=900 + i, out=out)
            # a timed region of >= 1 s (3 steps were 0.23 s): calibrate on one step, then time n_ex
            _sync(device)
            t1 = time.perf_counter()
            ex.plan(z0, disc, pe, warm, task_emb=emb, act_mask=mask, seed=902, out=out)
            _sync(device)
            n_ex = max(3, int(np.ceil(1.1 / max(time.perf_counter() - t1, 1e-3))))
            ex.set_profiling(n_ex * I)
            _sync(device)
            t1 = time.perf_counter()
            for i in range(n_ex):
                ex.plan(z0, disc, pe, warm, task_emb=emb, act_mask=mask, seed=910 + i, out=out)
            _sync(device)
            el = time.perf_counter() - t1
            ms, n = ex.profile_read()
            ach = flops_rollout_launch(cfg, E) / (ms / 1e3 / max(n, 1)) / 1e12
            ach_x = flops_rollout_executed(cfg, E, family == "fused") / (ms / 1e3 / max(n, 1)) / 1e12
            extra["exact_fp32_mode"] = {
                "value": round(n_ex * E / el, 2), "unit": "plans/s (this rank)", "steps": n_ex, "seconds": round(el, 3),
                "arithmetic": "fp32 MFMA (v_mfma_f32_32x32x2_f32): bitwise an fmaf chain",
                "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                             # `frac` prices what the kernel EXECUTES (2 of num_q heads, shared z0 products): <= 1 by construction;
                             # the as-written figure (all num_q heads) exceeds the roof because 3 of 5 heads are never computed
                             "frac": round(ach_x / FP32_MFMA_PEAK_TFLOPS, 4), "achieved_executed": round(ach_x, 2),
                             "frac_as_written": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                             "avg_launch_ms": round(ms / max(n, 1), 4)}}
            ex.close()

        if args.config == "c2" and world == 1 and not args.skip_extra_configs:
            planner.close()  # free the c2 workspace before the 317M model arrives
            extra["configs"] = {}
            # timed regions of >= 1 s each.  c2_i8: the headline model at the reference's OWN iteration count (tdmpc2.py:34: + 2 for
            # action_dim >= 20); c4_l1024: BASELINE configs[3] as literally written (latent_dim 1024; the reference's 317M has 1376 = c4)
            # c5_share: BASELINE configs[4]'s PER-GPU share (64 of the 512 envs of the 317M model) on this one GPU -- the N = 1 point the
            # first SCALE record's c5 leg (c5_leg, N > 1 only) divides by
            for key, name, e_leg, k_leg in (("c3", "c3", 30, 44), ("c4", "c4", 8, 13), ("c2_i8", "c2", 256, 32), ("c4_l1024", "c4_l1024", 8, 13),
                                            ("c5_share", "c4", 64, 2)):
                try:
                    extra["configs"][key] = config_leg(name, e_leg, k_leg, device, rank, single_env=key in ("c3", "c4"))
                    log(f"extra config {key}: {extra['configs'][key]['value']} plans/s")
                except Exception as ex:
                    extra["configs"][key] = {"error": repr(ex)}

    c5 = None
    # (TDMPC2_BENCH_FORCE_C5=1 under torch.distributed.run exercises the leg with one rank: the builder's pool has one GPU)
    if (world > 1 or (use_dist and os.environ.get("TDMPC2_BENCH_FORCE_C5"))) and args.config == "c2" and not args.skip_extra_configs:
        try:
            planner.close()
            c5 = c5_leg(device, rank, world, fence)
        except Exception as ex:
            c5 = {"error": repr(ex)}
    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    if c5 is not None:
        extra.setdefault("configs", {})["c5"] = c5

    plans = world * E * K
    value = plans / elapsed
    launch_s = (roll_ms / 1e3) / max(roll_n, 1)
    achieved = flops_rollout_launch(cfg, E) / launch_s / 1e12
    executed = flops_rollout_executed(cfg, E, family == "fused") / launch_s / 1e12
    split = planner.precision == 2
    kernel = ("ks_rollout" if family == "fused"
              else ("g_gemm_w / g_gemm_s" if split else "g_gemm") + " + row kernels of one _estimate_value")
    traffic, traffic_info = None, {"skipped": True}
    traffic = None if traffic is None else round(traffic)
    peak = F16_MFMA_PEAK_TFLOPS if split else FP32_MFMA_PEAK_TFLOPS
    line = {
        "metric": "plan() calls/sec (H=3, 512 samples, 6 iters)",
        "value": round(value, 2),
        "unit": "plans/s",
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": round(1e3 * elapsed / K, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        # inputs, outputs, accumulation and all non-GEMM math are fp32; what the label must not hide is how the PRODUCTS of the
        # nn.Linear contractions are formed (extra.exact_fp32_mode carries the strict-fp32 figure of the same run)
        "dtype": "f32 (f16x2-split products: 22-bit operands, fp32 accumulate)" if split else "f32",
        "data": "synthetic",
        "config": {
            "workload": f"{args.config}: {cfg.task} world model (L{cfg.latent_dim} M{cfg.mlp_dim} A{cfg.action_dim} "
                        f"nq{cfg.num_q} T{cfg.task_dim}), plan() H={cfg.horizon} N={cfg.num_samples} K={cfg.num_elites} "
                        f"P={cfg.num_pi_trajs} I={I}, {E} independent envs per GPU per step, random-init weights, "
                        f"SimNorm latents, in-kernel Philox noise",
            "envs_per_gpu": E, "iterations": I, "parallelism": f"env-sharded x{world}", "kernel_family": family, "arithmetic": arith,
            "gflop_per_plan_as_written": round(flops_plan(cfg, I) / 1e9, 3),
        },
        "roofline": {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(achieved / peak, 4), "traffic": traffic,
            "achieved_executed": round(executed, 2),
            "mfma_issue_executed": round(executed * (3 if split else 1), 2),
            "frac_mfma_issue_executed": round(executed * (3 if split else 1) / peak, 4),
            "traffic_unit": "bytes per launch (HBM/fabric side of L2: FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)",
            "traffic_detail": traffic_info,
            "kernel": kernel,
            "launches_timed": roll_n, "avg_launch_ms": round(1e3 * launch_s, 4),
            "note": "achieved = ALGORITHMIC (as-written) FLOPs of one CEM iteration (all num_q Q heads, SURVEY 8(d)) x envs / "
                    "mean duration of that iteration's rollout stage (HIP events on the launch stream); the kernels "
                    "execute fewer algorithmic FLOPs (2 of num_q heads; fused: shared z0 product at t=0): achieved_executed; "
                    "mfma_issue_executed = executed x MFMA products per algorithmic product, the figure to hold against "
                    "the PMC matrix-pipe utilisation in profiles/"
                    + ("; peak = dense f16 MFMA; the f16x2-split arithmetic spends 3 MFMA FLOPs per algorithmic FLOP "
                       "(a_hi.b_hi + a_hi.b_lo + a_lo.b_hi, fp32 accumulate, fp32-class error), so the ceiling of this "
                       "arithmetic in algorithmic FLOPs is peak/3" if split else "; peak = fp32-input MFMA (exact fp32)"),
        },
        "plan_tflops_as_written": round(value * flops_plan(cfg, I) / 1e12, 2),
        **({"roofline_split_arithmetic": {"peak_algorithmic": round(F16_MFMA_PEAK_TFLOPS / 3, 1), "unit": "TFLOP/s",
                                          "frac": round(achieved / (F16_MFMA_PEAK_TFLOPS / 3), 4),
                                          "vs_fp32_mfma_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 3)}} if split else {}),
        "extra": extra,
    }
    if world == 1 and not args.skip_cpu_baseline and not STUB:
        try:
            line["extra"]["torch_rocm_eager_same_gpu"] = torch_eager_gpu_baseline(cfg, I, sd_np, device)
        except Exception as ex:
            line["extra"]["torch_rocm_eager_same_gpu"] = {"error": repr(ex)}
        if args.config in ("c1", "c2"):
            log("torch.compile(reduce-overhead) baseline (child process)")
            line["extra"]["torch_compile_same_gpu"] = torch_compile_baseline(args.config, I)
        try:
            line["cpu_baseline"] = cpu_baseline(cfg, I, sd_np, args.cpu_budget)
        except Exception as ex:  # the baseline is a reported number, never a reason to lose the measurement
            line["cpu_baseline"] = {"value": None, "error": repr(ex)}
        if args.config == "c2" and not args.skip_extra_configs:
            # BASELINE.json configs[0] -- the reference's own CPU-runnable case (cheetah-run 5M, plan() on CPU PyTorch, H3 N512 I6): the
            # same CPU restatement on c1's dimensions, a few seconds (a reported baseline like cpu_baseline, nothing of the product)
            try:
                c1 = named_config("c1")
                leg = cpu_baseline(c1, 6, synth.make_state_dict(c1, seed=0), budget_s=4.0)
                leg["config"] = {"workload": "c1: cheetah-run 5M (L512 M512 A6 nq5), plan() H=3 N=512 I=6 on the host CPU (torch), 1 env"}
                line["extra"].setdefault("configs", {})["c1_cpu"] = leg
            except Exception as ex:
                line["extra"].setdefault("configs", {})["c1_cpu"] = {"error": repr(ex)}
    # last (a counter pass that times out cannot cost the baselines above): roofline.traffic from rocprofv3 --pmc children
    if world == 1 and not args.skip_traffic and args.path == "auto" and args.precision == "auto" and not STUB:
        log("measuring roofline.traffic (rocprofv3 --pmc children)")
        traffic, traffic_info = measured_traffic(args.config, E, I, family == "fused", E * cfg.num_samples // 64)
        for nm, leg in extra.get("configs", {}).items():
            if nm in ("c3", "c4") and "roofline" in leg:
                tb, ti = measured_traffic(nm, leg["config"]["envs"], leg["config"]["iterations"], leg["config"]["kernel_family"] == "fused",
                                          leg["config"]["envs"] * named_config(nm).num_samples // 64)
                leg["roofline"]["traffic"] = None if tb is None else round(tb)
                leg["roofline"]["traffic_detail"] = ti
        line["roofline"]["traffic"] = None if traffic is None else round(traffic)
        line["roofline"]["traffic_detail"] = traffic_info
    print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
