"""Golden-case table shared by `oracle/make_golden.py` (generator, build
container only) and the tests (consumers, anywhere).  TEST INFRASTRUCTURE ONLY.

A case is fully determined by (config name + overrides, seeds): inputs are
rebuilt with `tdmpc2_amd.synth`; only the reference's OUTPUTS are stored in
`tests/golden/<case>.npz`.
"""
from __future__ import annotations

import numpy as np
import torch

from tdmpc2_amd import synth
from tdmpc2_amd.config import get_discount, named_config, planner_iterations

CASES = {
    # name: (config, overrides, n_envs, eval_mode, head_std)
    "tiny": ("tiny", {}, 3, False, 0.06),
    "tiny_eval": ("tiny", {}, 2, True, 0.06),
    "tiny_mt": ("tiny", dict(task="mt30", model_size=None), 3, False, 0.06),
    "c1": ("c1", {}, 2, False, 0.06),
    # more plans per fixture: eight environments of the 5M model, the reference's 6 iterations each
    "c1_x8": ("c1", {}, 8, False, 0.06),
    "c1_wide": ("c1", {}, 1, False, 0.1),
    "c2": ("c2", {}, 2, False, 0.06),
    "mt5": ("mt5", {}, 2, False, 0.06),
    # layered kernel family (any latent_dim / mlp_dim, episodic termination head)
    "small": ("small", {}, 3, False, 0.06),
    "small_ep": ("small", dict(episodic=True), 2, False, 0.06),
    "small_mt": ("small", dict(task="mt30"), 3, False, 0.06),
    # the same episodic model with a termination head that FIRES (a tenth to a half of the sample rows per step; with the plain
    # synthetic weights of "small_ep" sigmoid(.) never crosses 0.5, so the layered family's (1 - term) masking was idle there)
    "small_ep_fire": ("small", dict(episodic=True), 2, False, 0.06),
    "c1_ep": ("c1", dict(episodic=True), 1, False, 0.06),
    "c3": ("c3", {}, 2, False, 0.03),                   # mt30 48M: L768 M1792 T64
    "c4": ("c4", dict(iterations=2), 1, False, 0.02),   # mt80 317M: L1376 M4096 nq8 T96, H5 N1024 (2 CEM iterations)
    # BASELINE.json's configs[3] as literally written: latent_dim=1024 (the reference's 317M uses 1376), H5 N1024
    "c4_l1024": ("c4_l1024", dict(iterations=2), 1, False, 0.02),
    # the benched setting of configs[1]: I = 6 CEM iterations (BASELINE.json metric text; the reference's own rule gives
    # 8 for action_dim >= 20, tdmpc2.py:34, which is what "c2" pins): base iterations 4 + 2
    "c2_i6": ("c2", dict(iterations=4), 2, False, 0.06),
    # episodic (termination head) at the dog-run dims: A = 38 exercises the 48-column action padding of the fused family
    "c2_ep": ("c2", dict(iterations=2, episodic=True), 1, False, 0.06),
    # fatter pins of the large models: the benched c4 leg runs 6 CEM iterations, "c4" pins two of them
    "c3_x4": ("c3", {}, 4, False, 0.03),                # mt30 48M, four plans (tasks 3, 10, 17, 24), 6 iterations
    "c4_x2": ("c4", {}, 2, False, 0.02),                # mt80 317M, two plans x the full 6 iterations, H5 N1024
    # the two model sizes the reference ships checkpoints for that had no fixture (VERDICT r4 missing #2)
    "m19_mt80": ("m19_mt80", {}, 2, False, 0.04),       # 19M: L768 M1024 T96 nq5, 6 iterations
    "m19_mt30": ("m19_mt30", {}, 2, False, 0.04),       # 19M as mt30 ships it: L512 M1024 T64
    "m1_mt30": ("m1_mt30", {}, 3, False, 0.06),         # 1M: L128 M384 nq2 T96, full length (6 iterations, three plans)
    # the reference's regression heads (common/math.py:60-63, 76-79: one output column; two_hot_inv = identity at num_bins 0,
    # symexp at 1) through the reference's own planner -- its parser.py:59 divides by num_bins - 1, the code behind it does not
    "c1_nb0": ("c1", dict(num_bins=0, iterations=3), 2, False, 0.06),
    "small_nb1_ep": ("small", dict(num_bins=1, episodic=True), 2, False, 0.06),
    # TRAINED-LIKE weight statistics (synth.trained_like: LayerNorm gains in [0.2, 5], biases N(0, 0.3), 0.1 % of every matrix at
    # 20 sigma) and head weights scaled for trajectory values of a few hundred (+- 300 .. 1000) -- the benched model and the two large ones
    # (VERDICT r5 next #4a: the split arithmetic's margin had only been shown on kind synthetic weights)
    "c2_tl": ("c2", dict(iterations=4), 2, False, 0.015),     # dog-run 5M at the benched I = 6
    "c3_tl": ("c3", {}, 2, False, 0.02),                    # mt30 48M
    "c4_tl": ("c4", dict(iterations=2), 1, False, 0.012),    # mt80 317M, H5 N1024, 2 iterations
}


def build_case(name: str):
    cfg_name, overrides, E, eval_mode, head_std = CASES[name]
    return build_custom(named_config(cfg_name, **overrides), E, eval_mode, head_std, name=name)


def build_custom(cfg, E: int, eval_mode: bool = False, head_std: float = 0.06, name: str = "", t0=None):
    """A case from an arbitrary config (edge-case tests build these on the fly; no golden fixture)."""
    if cfg.multitask and name in ("tiny_mt", "small_mt", "c3", "c4", "c4_l1024", "c3_x4", "c4_x2", "m19_mt80", "m19_mt30", "m1_mt30", "c3_tl", "c4_tl"):
        # heterogeneous action dims / episode lengths to exercise masks and per-task discounts
        n = len(cfg.tasks)
        cfg.action_dims = [cfg.action_dim - (i % 3) for i in range(n)]
        cfg.episode_lengths = [500 if i % 2 == 0 else 100 for i in range(n)]
    if cfg.multitask and name == "mt5":
        n = len(cfg.tasks)
        cfg.action_dims = [cfg.action_dim - (i % 4) for i in range(n)]
        cfg.episode_lengths = [500 if i % 2 == 0 else 1000 for i in range(n)]
    I = planner_iterations(cfg)
    sd = synth.make_state_dict(cfg, seed=0, head_std=head_std)
    if name.endswith("_tl"):
        sd = synth.trained_like(sd, seed=0)
    if name == "small_ep_fire":  # spread the termination logits around 0 (world_model.py:132-141: sigmoid(.) > 0.5 terminates)
        sd["_termination.2.weight"] = sd["_termination.2.weight"] * 12.0
        sd["_termination.2.bias"] = sd["_termination.2.bias"] * 0.0 + 3.3
    z0 = synth.make_latents(cfg, E, seed=1)
    tape = synth.make_noise_tape(cfg, E, I, seed=2)
    prev = np.random.default_rng(5).uniform(-0.5, 0.5, (E, cfg.horizon, cfg.action_dim)).astype(np.float32)
    t0 = np.array([(e % 2 == 0) for e in range(E)]) if t0 is None else np.asarray(t0, dtype=bool)
    if cfg.multitask:
        tasks = [(7 * e + 3) % len(cfg.tasks) for e in range(E)]
        disc_t = torch.tensor([get_discount(cfg, L) for L in cfg.episode_lengths])  # tdmpc2.py:35-37
        discounts = [disc_t[t] for t in tasks]
    else:
        tasks = None
        discounts = [get_discount(cfg, cfg.episode_length)] * E
    return dict(cfg=cfg, iterations=I, sd=sd, z0=z0, tape=tape, prev_mean=prev, t0=t0, tasks=tasks,
                discounts=discounts, eval_mode=eval_mode, n_envs=E)


def td_batch(cfg, B: int = 40, seed: int = 5):
    """Synthetic inputs of one `TDMPC2._td_target` call (tdmpc2/tdmpc2.py:239-254): next_z [H, B, L] SimNorm latents,
    reward [H, B, 1], terminated [H, B, 1] in {0, 1}, the policy's noise and the two Q heads."""
    rng = np.random.default_rng(seed)
    H = cfg.horizon
    z = synth.make_latents(cfg, H * B, seed=seed).reshape(H, B, cfg.latent_dim)
    tasks = (np.arange(B) % len(cfg.tasks)).astype(np.int64) if cfg.multitask else None  # task [B] as in tdmpc2.py:249
    return {"tasks": tasks, "next_z": z, "reward": rng.standard_normal((H, B, 1)).astype(np.float32),
            "terminated": (rng.random((H, B, 1)) < 0.2).astype(np.float32),
            "pi_eps": rng.standard_normal((H, B, cfg.action_dim)).astype(np.float32),
            "qidx": np.array([min(3, cfg.num_q - 1), 1], np.int32) if cfg.num_q > 2 else np.array([1, 0], np.int32)}
