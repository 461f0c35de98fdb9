"""-m gpu: the HIP state encoder (tdmpc2_plan_bind_encoder / tdmpc2_plan_encode / tdmpc2_plan_run_obs)
against the reference's own `WorldModel.encode` output (tests/golden/*.npz: `encode_z`) and against the oracle."""
import numpy as np
import pytest
import torch

from tests.gpu_common import case_on_gpu, dev, plan_inputs
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu

ENC_ATOL = 2e-6  # SimNorm outputs are in [0, 1]; fp32 summation order is the only difference


def _bind(c, model, planner):
    planner.bind_encoder({k: v for k, v in model.sd.items() if k.startswith("_encoder.state.")})


def _emb(c, model):
    if not c["cfg"].multitask:
        return None
    return plan_inputs(c, model)["task_emb"]


@pytest.mark.parametrize("name", ["small", "small_mt", "c1", "c2", "mt5"])
def test_encode_matches_reference_golden(name):
    from tdmpc2_amd import synth

    c, model, planner = case_on_gpu(name)
    _bind(c, model, planner)
    g = load_golden(name)
    obs = torch.as_tensor(synth.make_obs(c["cfg"], c["n_envs"], seed=3)).to(dev())
    z = planner.encode(obs, _emb(c, model)).cpu().numpy()
    assert z.shape == g["encode_z"].shape
    err = np.abs(z - g["encode_z"]).max()
    print(f"[{name}] encoder max |HIP - reference| = {err:.2e}")
    assert err < ENC_ATOL
    # SimNorm property (layers.py:74-91): every group of simnorm_dim latents sums to one
    np.testing.assert_allclose(z.reshape(z.shape[0], -1, c["cfg"].simnorm_dim).sum(-1), 1.0, atol=1e-5)


@pytest.mark.parametrize("layers,enc_dim,latent,envs", [(1, 96, 64, 3), (3, 200, 128, 5), (5, 1024, 1376, 2), (4, 4096, 512, 1)])
def test_encode_depths_and_widths_against_oracle(layers, enc_dim, latent, envs):
    """num_enc_layers 1..5 (common/__init__.py:1-24), widths up to the 317M model's 4096, latent 1376 = 172 SimNorm groups."""
    from oracle import planner_oracle as po
    from tdmpc2_amd import synth
    from tdmpc2_amd.config import named_config
    from tdmpc2_amd.native import NativePlanner

    cfg = named_config("small", num_enc_layers=layers, enc_dim=enc_dim, latent_dim=latent, mlp_dim=64)
    sd = {k: torch.as_tensor(v) for k, v in synth.make_state_dict(cfg, seed=layers).items()}
    model = po.OracleModel(cfg, sd)
    planner = NativePlanner(cfg, cfg.iterations, dev(), max_envs=envs)
    planner.bind_encoder({k: v for k, v in sd.items() if k.startswith("_encoder.state.")})
    assert planner.encoder_layers == max(layers - 1, 1) + 1
    obs = torch.as_tensor(synth.make_obs(cfg, envs, seed=11))
    want = torch.cat([model.encode(obs[e:e + 1], None) for e in range(envs)]).numpy()
    z = planner.encode(obs.to(dev())).cpu().numpy()
    err = np.abs(z - want).max()
    print(f"[enc {layers} layers, {enc_dim} wide, L={latent}] max |HIP - oracle| = {err:.2e}")
    assert err < ENC_ATOL


@pytest.mark.parametrize("name", ["c1", "mt5"])
def test_run_obs_equals_encode_then_run(name):
    """tdmpc2_plan_run_obs = tdmpc2_plan_encode + tdmpc2_plan_run (same tape): bit-identical actions and prev_mean."""
    from tdmpc2_amd import synth

    c, model, planner = case_on_gpu(name)
    _bind(c, model, planner)
    inp = plan_inputs(c, model)
    obs = torch.as_tensor(synth.make_obs(c["cfg"], c["n_envs"], seed=3)).to(dev())
    z = planner.encode(obs, inp["task_emb"])
    pm1, pm2 = inp["prev_mean"].clone(), inp["prev_mean"].clone()
    a1 = planner.plan(z, inp["disc_pow"], pm1, inp["t0"], task_emb=inp["task_emb"], act_mask=inp["act_mask"], tape=inp["tape"])
    a2 = planner.plan_obs(obs, inp["disc_pow"], pm2, inp["t0"], task_emb=inp["task_emb"], act_mask=inp["act_mask"], tape=inp["tape"])
    assert torch.equal(a1, a2) and torch.equal(pm1, pm2)


def test_encoder_errors():
    from tdmpc2_amd.native import NativeError

    c, model, planner = case_on_gpu("small")
    sd = {k: v for k, v in model.sd.items() if k.startswith("_encoder.state.")}
    from tdmpc2_amd.native import NativePlanner

    fresh = NativePlanner(c["cfg"], c["iterations"], dev(), max_envs=2)
    fresh.obs_dim = c["cfg"].obs_shape["state"][0]
    with pytest.raises(NativeError, match="no encoder bound"):
        fresh.encode(torch.zeros(1, fresh.obs_dim, device=dev()))
    bad = dict(sd)
    last = max(int(k.split(".")[2]) for k in sd)
    bad[f"_encoder.state.{last}.weight"] = sd[f"_encoder.state.{last}.weight"][:-8]
    for n in ("bias", "ln.weight", "ln.bias"):
        bad[f"_encoder.state.{last}.{n}"] = sd[f"_encoder.state.{last}.{n}"][:-8]
    with pytest.raises(NativeError, match="latent_dim"):
        fresh.bind_encoder(bad)
