"""Dependency-free planner configuration.

Mirrors the fields the reference planner reads from its hydra config
(reference: tdmpc2/config.yaml:33-64, tdmpc2/common/parser.py:29-80,
tdmpc2/common/__init__.py:1-60) without hydra/omegaconf.  Field names are the
reference's, so code written against ``cfg.horizon`` etc. keeps working.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

# reference: tdmpc2/common/__init__.py:1-24
MODEL_SIZE = {
    1: dict(enc_dim=256, mlp_dim=384, latent_dim=128, num_enc_layers=2, num_q=2),
    5: dict(enc_dim=256, mlp_dim=512, latent_dim=512, num_enc_layers=2),
    19: dict(enc_dim=1024, mlp_dim=1024, latent_dim=768, num_enc_layers=3),
    48: dict(enc_dim=1792, mlp_dim=1792, latent_dim=768, num_enc_layers=4),
    317: dict(enc_dim=4096, mlp_dim=4096, latent_dim=1376, num_enc_layers=5, num_q=8),
}

# reference: tdmpc2/common/__init__.py:26-60 (only the sizes matter to the planner)
TASK_SET_SIZE = {"mt30": 30, "mt80": 80}


@dataclass
class Config:
    # environment
    task: str = "cheetah-run"
    obs: str = "state"
    episodic: bool = False
    # planning (config.yaml:33-42)
    mpc: bool = True
    iterations: int = 6
    num_samples: int = 512
    num_elites: int = 64
    num_pi_trajs: int = 24
    horizon: int = 3
    min_std: float = 0.05
    max_std: float = 2.0
    temperature: float = 0.5
    # actor (config.yaml:44-47)
    log_std_min: float = -10.0
    log_std_max: float = 2.0
    # critic (config.yaml:49-52)
    num_bins: int = 101
    vmin: float = -10.0
    vmax: float = 10.0
    # architecture (config.yaml:54-64)
    model_size: Optional[int] = 5
    num_enc_layers: int = 2
    enc_dim: int = 256
    num_channels: int = 32
    mlp_dim: int = 512
    latent_dim: int = 512
    task_dim: int = 0
    num_q: int = 5
    dropout: float = 0.01
    simnorm_dim: int = 8
    # discount heuristic (config.yaml:26-28)
    discount_denom: float = 5.0
    discount_min: float = 0.95
    discount_max: float = 0.995
    # misc
    compile: bool = False
    seed: int = 1
    # filled by the environment in the reference (envs/__init__.py:76-82)
    multitask: bool = False
    tasks: List[str] = field(default_factory=lambda: ["cheetah-run"])
    obs_shape: Dict[str, Tuple[int, ...]] = field(default_factory=lambda: {"state": (17,)})
    action_dim: int = 6
    episode_length: int = 500
    obs_shapes: Optional[List[Tuple[int, ...]]] = None
    action_dims: Optional[List[int]] = None
    episode_lengths: Optional[List[int]] = None
    bin_size: float = 0.2

    def replace(self, **kw) -> "Config":
        return dataclasses.replace(self, **kw)


def parse_cfg(cfg: Config) -> Config:
    """Apply the reference's derivations (parser.py:59-78) to a Config."""
    cfg = dataclasses.replace(cfg)
    cfg.bin_size = (cfg.vmax - cfg.vmin) / (cfg.num_bins - 1) if cfg.num_bins != 1 else 0.0  # parser.py:59 (which divides by zero at 1)
    if cfg.model_size is not None:  # parser.py:62-68
        if cfg.model_size not in MODEL_SIZE:
            raise ValueError(f"Invalid model size {cfg.model_size}. Must be one of {list(MODEL_SIZE)}")
        for k, v in MODEL_SIZE[cfg.model_size].items():
            setattr(cfg, k, v)
        if cfg.task == "mt30" and cfg.model_size == 19:
            cfg.latent_dim = 512
    cfg.multitask = cfg.task in TASK_SET_SIZE  # parser.py:71
    if cfg.multitask:  # parser.py:75
        ms = cfg.model_size if cfg.model_size is not None else 5
        cfg.task_dim = 96 if (cfg.task == "mt80" or ms in {1, 317}) else 64
        n = TASK_SET_SIZE[cfg.task]
        cfg.tasks = [f"{cfg.task}-{i}" for i in range(n)]
        if cfg.action_dims is None:
            cfg.action_dims = [cfg.action_dim] * n
        if cfg.episode_lengths is None:
            cfg.episode_lengths = [cfg.episode_length] * n
    else:
        cfg.task_dim = 0
        cfg.tasks = [cfg.task]
    return cfg


def get_discount(cfg: Config, episode_length: int) -> float:
    """reference: tdmpc2/tdmpc2.py:57-70."""
    frac = episode_length / cfg.discount_denom
    return min(max((frac - 1) / frac, cfg.discount_min), cfg.discount_max)


def planner_iterations(cfg: Config) -> int:
    """reference: tdmpc2/tdmpc2.py:34 (+2 iterations for large action spaces)."""
    return cfg.iterations + 2 * int(cfg.action_dim >= 20)


# ---- BASELINE.json configurations (dims marked with a dagger
# there come from the simulators and are reproduced as constants here) ----
def named_config(name: str, **overrides) -> Config:
    if name == "c1":  # cheetah-run 5M
        cfg = Config(task="cheetah-run", model_size=5, action_dim=6, obs_shape={"state": (17,)}, episode_length=500)
    elif name == "c2":  # dog-run 5M
        cfg = Config(task="dog-run", model_size=5, action_dim=38, obs_shape={"state": (223,)}, episode_length=500)
    elif name == "c3":  # mt30 48M
        cfg = Config(task="mt30", model_size=48, action_dim=6, obs_shape={"state": (24,)}, episode_length=500)
    elif name == "c4":  # mt80 317M, H5 N1024
        cfg = Config(task="mt80", model_size=317, action_dim=6, obs_shape={"state": (39,)}, episode_length=500,
                     horizon=5, num_samples=1024)
    elif name == "mt5":  # multitask at the 5M dims (fused-kernel size class)
        cfg = Config(task="mt30", model_size=5, action_dim=6, obs_shape={"state": (24,)}, episode_length=500)
    elif name == "c4_l1024":  # BASELINE.json's text for configs[3] says latent_dim=1024 (not the reference's 1376)
        cfg = Config(task="mt80", model_size=317, action_dim=6, obs_shape={"state": (39,)}, episode_length=500,
                     horizon=5, num_samples=1024)
        cfg = parse_cfg(cfg)
        cfg.latent_dim = 1024
        for k, v in overrides.items():
            setattr(cfg, k, v)
        return cfg
    elif name == "m19_mt80":  # the 19M multitask checkpoint class (common/__init__.py:11-14): L768 M1024 T96
        cfg = Config(task="mt80", model_size=19, action_dim=6, obs_shape={"state": (39,)}, episode_length=500)
    elif name == "m19_mt30":  # mt30's 19M checkpoint is "slightly smaller" (parser.py:67-68): L512 M1024 T64
        cfg = Config(task="mt30", model_size=19, action_dim=6, obs_shape={"state": (24,)}, episode_length=500)
    elif name == "m1_mt30":  # the 1M multitask class: L128 M384 nq2, task_dim 96 (parser.py:75)
        cfg = Config(task="mt30", model_size=1, action_dim=6, obs_shape={"state": (24,)}, episode_length=500)
    elif name == "small":  # small dims on the layered kernel family's tiling (num_samples % 128 == 0, dims % 32 == 0)
        cfg = Config(task="walker-run", model_size=None, latent_dim=64, mlp_dim=96, enc_dim=32, num_q=3,
                     action_dim=5, obs_shape={"state": (11,)}, num_samples=128, num_elites=16, num_pi_trajs=8,
                     horizon=3, iterations=3)
    elif name == "tiny":  # small dims for fast CPU tests of the oracle / host logic
        cfg = Config(task="cheetah-run", model_size=None, latent_dim=64, mlp_dim=64, enc_dim=32, num_q=3,
                     action_dim=4, obs_shape={"state": (9,)}, num_samples=64, num_elites=8, num_pi_trajs=8,
                     horizon=2, iterations=3)
    else:
        raise KeyError(name)
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return parse_cfg(cfg)
