"""-m gpu: an evaluate.py-shaped episode loop (reference tdmpc2/evaluate.py:71-85) through the drop-in class, against the oracle
step by step.

    for task_idx, task in enumerate(tasks):            # multitask: one task after the other on ONE agent
        for i in range(eval_episodes):
            obs, done, t = env.reset(task_idx), False, 0
            while not done:
                action = agent.act(obs, t0=t==0, task=task_idx)      # eval_mode defaults to False: the final noise is added
                obs, reward, done, info = env.step(action)
                t += 1

What two consecutive steps (tests/test_gpu_boundary.py) cannot show: the t0 re-arm at every episode start, `_prev_mean` carried
over dozens of warm starts, and the task switch of the multitask loop (per-task discount, action mask, embedding) on one handle.
The HIP side plans with its in-kernel Philox generator (tape = NULL, what a deployment runs); every call's draws are exported
(tdmpc2_plan_export_noise) and replayed through the oracle, which keeps its OWN `_prev_mean` chain.  The environment is a fixed
random contraction obs' = tanh(W obs + B a + c): deterministic, so both sides see the same observations."""
import numpy as np
import pytest
import torch

from tests.helpers import ACT_ATOL, boundary_gap, record_parity

pytestmark = pytest.mark.gpu


class ToyEnv:
    """obs' = tanh(W obs + B a + c); an episode ends after `length` steps."""

    def __init__(self, obs_dim, act_dim, length, seed=0):
        rng = np.random.default_rng(seed)
        self.W = (rng.standard_normal((obs_dim, obs_dim)) / np.sqrt(obs_dim)).astype(np.float32)
        self.B = rng.standard_normal((obs_dim, act_dim)).astype(np.float32) * 0.5
        self.c = rng.standard_normal(obs_dim).astype(np.float32) * 0.1
        self.rng, self.length, self.obs_dim = rng, length, obs_dim

    def reset(self, task_idx=None):
        self.t = 0
        self.obs = self.rng.standard_normal(self.obs_dim).astype(np.float32) * (1.0 + 0.1 * (task_idx or 0))
        return torch.as_tensor(self.obs)

    def step(self, action):
        self.obs = np.tanh(self.W @ self.obs + self.B @ action.numpy().astype(np.float32) + self.c).astype(np.float32)
        self.t += 1
        return torch.as_tensor(self.obs), 0.0, self.t >= self.length, {}


@pytest.mark.parametrize("name,episodes,length,n_tasks", [("c1", 3, 25, 1), ("mt5", 1, 12, 3), ("small_mt", 2, 10, 3)])
def test_evaluate_shaped_episode_loop_matches_the_oracle_step_by_step(name, episodes, length, n_tasks):
    from oracle import cases
    from oracle import planner_oracle as po
    from tdmpc2_amd.tdmpc2 import TDMPC2

    c = cases.build_case(name)
    cfg = c["cfg"]
    agent = TDMPC2(cfg.replace(), device=torch.device("cuda", 0))
    agent.load({"model": {k: torch.as_tensor(v) for k, v in c["sd"].items()}})
    model = po.OracleModel(cfg, {k: torch.as_tensor(v) for k, v in c["sd"].items()})
    planner = agent.planner()
    I = agent.cfg.iterations
    obs_dim = list(cfg.obs_shape.values())[0][0]
    env = ToyEnv(obs_dim, cfg.action_dim, length, seed=11)
    tasks = list(range(n_tasks)) if cfg.multitask else [None]
    o_prev = torch.zeros(cfg.horizon, cfg.action_dim)  # the oracle's own _prev_mean buffer (tdmpc2.py:40)
    worst_a = worst_pm = 0.0
    steps = swaps = 0
    for task_idx in tasks:
        if cfg.multitask:
            disc = model_discount = agent.discount[task_idx].cpu()
        else:
            disc = agent.discount
        for ep in range(episodes):
            obs, done, t = env.reset(task_idx), False, 0
            while not done:
                seed, call = agent._seed + 1, planner.call_counter()  # what the coming act() will plan under
                a = agent.act(obs, t0=t == 0, task=task_idx)
                assert a.device.type == "cpu" and a.shape == (cfg.action_dim,) and torch.isfinite(a).all() and a.abs().max() <= 1
                assert planner.call_counter() == call + 1 and agent._seed == seed  # no silent re-plan
                tape = {k: v[0].cpu() for k, v in planner.export_noise(seed, call, 1).items()}
                wa, wpm, st = po.plan(model, obs=obs.unsqueeze(0), tape=tape, prev_mean=o_prev, t0=t == 0, eval_mode=False,
                                      task=task_idx, discount=disc, iterations=I)
                da = (a - wa).abs().max().item()
                dm = (agent._prev_mean.cpu() - wpm).abs().max().item()
                steps += 1
                if da >= ACT_ATOL or dm >= ACT_ATOL:
                    # legitimate only through an elite-boundary swap (top-k is discontinuous where the reference's own k-th and
                    # (k+1)-th values are closer than 1e-4); counted, and the oracle's chain restarts from the planner's buffer
                    gaps = [boundary_gap(st["value"][it].numpy(), cfg.num_elites) for it in range(I)]
                    assert min(gaps) < 1e-4, (f"{name}: task {task_idx} episode {ep} step {t}: action diff {da:.2e}, prev_mean diff "
                                              f"{dm:.2e} with no elite boundary within 1e-4 (gaps {gaps})")
                    swaps += 1
                    o_prev = agent._prev_mean.cpu().clone()
                else:
                    worst_a, worst_pm = max(worst_a, da), max(worst_pm, dm)
                    o_prev = wpm
                if cfg.multitask:  # the action mask of THIS task (a stale mask after a task switch would show here)
                    mask = model.sd["_action_masks"][task_idx]
                    assert torch.all(a[mask == 0] == 0)
                obs, _, done, _ = env.step(a)
                t += 1
    print(f"[{name}] {steps} act() calls over {len(tasks)} task(s) x {episodes} episode(s): worst action diff {worst_a:.2e}, "
          f"worst _prev_mean diff {worst_pm:.2e}, elite-boundary swaps {swaps}")
    record_parity(f"{name}/act()/episode_loop", action_abs=worst_a, prev_mean_abs=worst_pm, elite_swaps=int(swaps), plans=int(steps))
    assert swaps <= max(1, steps // 20)
    assert planner.take_fault() == 0
