"""-m gpu: edge configurations of plan() against the oracle on the same seeded inputs (no golden fixture: the
oracle itself is pinned against the reference in tests/test_oracle_golden.py): minimum / maximum sizes of the
kernels' envelope, no policy-prior trajectories, every sample an elite, horizon 1, eval mode, all-warm starts,
two-member Q ensembles, odd action widths, episodic planning at horizon 1, sample counts that are not tile multiples."""
import numpy as np
import pytest
import torch

from tests.test_gpu_planner import _compare_stages, _run_native

pytestmark = pytest.mark.gpu


def _edge_cases():
    from tdmpc2_amd.config import named_config

    return {
        # name: (cfg, E, eval_mode, t0, path, precision)
        "fused_min": (named_config("c1", horizon=1, num_samples=64, num_elites=64, num_pi_trajs=0, iterations=2), 3, False, None, 1, 2),
        "fused_min_fp32": (named_config("c1", horizon=1, num_samples=64, num_elites=64, num_pi_trajs=0, iterations=2), 2, False, None, 1, 1),
        "fused_max": (named_config("c1", horizon=5, num_samples=1024, num_elites=64, num_pi_trajs=64, iterations=2,
                                   action_dim=61), 1, False, None, 1, 2),
        "fused_eval_warm": (named_config("c1", iterations=3), 2, True, [False, False], 1, 2),
        "fused_nq2_one_elite": (named_config("c1", num_q=2, num_elites=1, iterations=3), 2, False, None, 1, 2),
        "fused_mt_odd_actions": (named_config("mt5", action_dim=17, iterations=2), 2, False, None, 1, 2),
        "layered_min": (named_config("small", horizon=1, num_elites=128, num_pi_trajs=0), 2, False, None, 2, 1),
        "layered_episodic_h1": (named_config("small", horizon=1, episodic=True), 2, True, [True, False], 2, 1),
        "layered_1m_model": (named_config("c1", model_size=1, task="mt30", action_dim=4, iterations=2), 2, False, None, 0, 0),
        "fused_episodic_h1_eval": (named_config("c1", horizon=1, episodic=True, iterations=2), 2, True, [True, False], 1, 2),
        "fused_episodic_h5_rows32": (named_config("c1", horizon=5, episodic=True, iterations=2, num_samples=128, num_elites=16), 1, False, None, 1, 1),
        # num_samples that are NOT multiples of the kernels' row tile (config.yaml:36 allows any value): NativePlanner creates the
        # handle with the count rounded up (512 / 256 / 128) and tdmpc2_plan_cfg::num_valid_samples = the true one -- the padding
        # rows are rolled out but never elites; values, elite sets, mean / std and the action are those of the oracle at the true count
        "fused_500_samples": (named_config("c1", num_samples=500, iterations=3), 2, False, None, 0, 0),
        "fused_500_samples_one_plan_cluster_path": (named_config("c1", num_samples=500, iterations=2), 1, False, None, 0, 0),
        "layered_200_samples": (named_config("small", num_samples=200, num_elites=16), 3, False, None, 0, 0),
        # the reference's regression heads (math.py:76-79): one output column, two_hot_inv = identity (num_bins 0) / symexp (1)
        "fused_num_bins_0": (named_config("c1", num_bins=0, iterations=2), 2, False, None, 0, 0),
        "fused_num_bins_1_one_plan": (named_config("c1", num_bins=1, iterations=2), 1, False, None, 0, 0),
        "layered_num_bins_0": (named_config("small", num_bins=0), 2, False, None, 0, 0),
        "layered_num_bins_1_episodic": (named_config("small", num_bins=1, episodic=True), 2, False, None, 0, 0),
        "layered_mt_72_samples_eval": (named_config("small", task="mt30", num_samples=72, num_elites=9, num_pi_trajs=5), 2, True, None, 0, 0),
    }


@pytest.mark.parametrize("name", list(_edge_cases()))
def test_edge_configuration_matches_oracle(name):
    from oracle import cases
    from oracle import planner_oracle as po
    from tdmpc2_amd.native import NativePlanner
    from tests.gpu_common import dev

    cfg, E, eval_mode, t0, path, prec = _edge_cases()[name]
    c = cases.build_custom(cfg, E, eval_mode=eval_mode, t0=t0)
    model = po.OracleModel(cfg, {k: torch.as_tensor(v) for k, v in c["sd"].items()})
    planner = NativePlanner(cfg, c["iterations"], dev(), max_envs=E, path=path, precision=prec)
    planner.bind_state_dict(model.sd)
    a, pm, st = po.plan_batch(model, c["z0"], c["tape"], c["prev_mean"], c["t0"], eval_mode, c["tasks"], c["discounts"],
                              c["iterations"])
    got = _run_native(c, model, planner)
    assert np.isfinite(got["action"]).all() and np.abs(got["action"]).max() <= 1.0
    _compare_stages(name, c, got, {k: v.numpy() for k, v in st.items()}, a.numpy(), pm.numpy(), tag="/edge/oracle")
    planner.close()
