"""CPU: host-side mirror of the reference interface — config derivations, checkpoint key layout,
discount powers, the encoder."""
import numpy as np
import pytest
import torch

from tdmpc2_amd import checkpoint, synth
from tdmpc2_amd.config import Config, get_discount, named_config, parse_cfg, planner_iterations
from tdmpc2_amd.world_model import WorldModel
from tests.helpers import load_golden


def test_model_size_table_and_task_dim_rules():
    # reference common/__init__.py:1-24, parser.py:59-78
    c = named_config("c1")
    assert (c.latent_dim, c.mlp_dim, c.num_q, c.task_dim, c.multitask) == (512, 512, 5, 0, False)
    assert abs(c.bin_size - 0.2) < 1e-12
    c3 = named_config("c3")
    assert (c3.latent_dim, c3.mlp_dim, c3.task_dim, len(c3.tasks)) == (768, 1792, 64, 30)
    c4 = named_config("c4")
    assert (c4.latent_dim, c4.mlp_dim, c4.num_q, c4.task_dim, len(c4.tasks)) == (1376, 4096, 8, 96, 80)
    assert parse_cfg(Config(task="mt30", model_size=19)).latent_dim == 512  # parser.py:67-68
    assert parse_cfg(Config(task="mt30", model_size=1)).task_dim == 96
    with pytest.raises(ValueError):
        parse_cfg(Config(model_size=7))


def test_iterations_and_discount_heuristics():
    # tdmpc2.py:34, 57-70
    assert planner_iterations(named_config("c1")) == 6
    assert planner_iterations(named_config("c2")) == 8
    cfg = named_config("c1")
    assert abs(get_discount(cfg, 500) - 0.99) < 1e-12
    assert get_discount(cfg, 25) == 0.95 and get_discount(cfg, 100000) == 0.995


def test_parameter_count_matches_model_name():
    wm = WorldModel(named_config("c1"))
    n = sum(p.numel() for p in wm.parameters())
    assert 4.9e6 < n < 5.0e6  # "5M"; SURVEY.md appendix B: 4.96 M


def test_state_dict_has_reference_keys():
    wm = WorldModel(named_config("mt5"))
    keys = set(wm.state_dict())
    for k in ["_encoder.state.0.weight", "_encoder.state.0.ln.bias", "_dynamics.2.ln.weight", "_reward.2.bias",
              "_pi.1.ln.weight", "_task_emb.weight", "_action_masks", "log_std_min", "log_std_dif",
              "_Qs.params.0.weight", "_Qs.params.1.ln.bias", "_Qs.params.2.bias", "_Qs.params.__batch_size",
              "_detach_Qs_params.0.weight", "_target_Qs_params.2.weight", "_target_Qs_params.__device"]:
        assert k in keys, k
    assert "_reward.2.ln.weight" not in keys and "_Qs.params.2.ln.weight" not in keys
    assert wm.state_dict()["_Qs.params.0.weight"].shape == (5, 512, 512 + 64 + 6)


def test_old_and_new_checkpoint_formats_load():
    cfg = named_config("tiny")
    wm = WorldModel(cfg)
    syn = {k: torch.as_tensor(v) for k, v in synth.make_state_dict(cfg, 0).items()}
    wm.load_state_dict(checkpoint.convert_state_dict(syn))
    new = wm.state_dict()
    old = checkpoint.to_old_format(new)
    assert "_Qs.params.9" in old and "_target_Qs.params.0" in old and "_detach_Qs_params.0.weight" not in old
    wm2 = WorldModel(cfg)
    wm2.load_state_dict(old)  # pre-hook converts
    for k, v in new.items():
        if torch.is_tensor(v):
            assert torch.equal(v, wm2.state_dict()[k]), k
    # mapping rule n = 4*layer + {weight, bias, ln.weight, ln.bias} (reference layers.py:175-193)
    assert torch.equal(old["_Qs.params.6"], new["_Qs.params.1.ln.weight"])


@pytest.mark.parametrize("name", ["tiny", "tiny_mt", "c1"])
def test_encoder_matches_reference_golden(name):
    from oracle import cases

    c = cases.build_case(name)
    wm = WorldModel(c["cfg"]).eval()
    wm.load_state_dict(checkpoint.convert_state_dict({k: torch.as_tensor(v) for k, v in c["sd"].items()}))
    g = load_golden(name)
    obs = synth.make_obs(c["cfg"], c["n_envs"], seed=3)
    with torch.no_grad():
        for e in range(c["n_envs"]):
            task = None if c["tasks"] is None else torch.tensor([c["tasks"][e]])
            z = wm.encode(torch.as_tensor(obs[e:e + 1]), task)[0].numpy()
            np.testing.assert_allclose(z, g["encode_z"][e], atol=1e-6, rtol=1e-5)


def test_host_world_model_matches_oracle_pieces():
    from oracle import cases
    from oracle import planner_oracle as po

    c = cases.build_case("tiny_mt")
    cfg = c["cfg"]
    sd = {k: torch.as_tensor(v) for k, v in c["sd"].items()}
    wm = WorldModel(cfg).eval()
    wm.load_state_dict(checkpoint.convert_state_dict(dict(sd)))
    om = po.OracleModel(cfg, sd)
    z = torch.as_tensor(c["z0"][:1]).repeat(5, 1)
    a = torch.rand(5, cfg.action_dim) * 2 - 1
    t = 4
    with torch.no_grad():
        tt = torch.tensor([t])
        assert torch.allclose(wm.next(z, a, tt), om.next(z, a, t), atol=1e-6)
        assert torch.allclose(wm.reward(z, a, tt), om.reward(z, a, t), atol=1e-5)
        qa = wm.Q(z, a, tt, return_type="all")
        qo = po.ensemble_forward(om.sd, "_Qs.params", torch.cat([om.task_emb(z, t), a], -1))
        assert torch.allclose(qa, qo, atol=1e-5)


def test_synthetic_inputs_are_deterministic():
    cfg = named_config("c1")
    a = synth.make_state_dict(cfg, 0)
    b = synth.make_state_dict(cfg, 0)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    z = synth.make_latents(cfg, 3, 1)
    assert np.allclose(z.reshape(3, -1, 8).sum(-1), 1.0, atol=1e-6) and (z >= 0).all()
    t = synth.make_noise_tape(cfg, 2, 6, 2)
    assert t["sample_eps"].shape == (2, 6, 3, 488, 6) and t["qidx"].shape == (2, 6, 2)
    assert (t["qidx"][..., 0] != t["qidx"][..., 1]).all()


def test_old_checkpoint_without_log_std_and_masks_loads():
    """Released old-API checkpoints carry neither log_std_min / log_std_dif nor _action_masks; the reference's loader
    takes them from the freshly built model (layers.py:211-215).  ADVICE r1: a strict load used to fail here."""
    cfg = named_config("tiny", task="mt30")
    wm = WorldModel(cfg)
    syn = {k: torch.as_tensor(v) for k, v in synth.make_state_dict(cfg, 0).items()}
    wm.load_state_dict(dict(syn))
    old = checkpoint.to_old_format(wm.state_dict())
    for k in ("log_std_min", "log_std_dif", "_action_masks"):
        old.pop(k)
    assert checkpoint.is_old_format(old) and not checkpoint.is_old_format(wm.state_dict())
    wm2 = WorldModel(cfg)
    wm2.load_state_dict(old)  # strict
    assert torch.equal(wm2._Qs.params.layer(2).bias, wm._Qs.params.layer(2).bias)
    assert float(wm2.log_std_min) == float(cfg.log_std_min)
    assert float(wm2.log_std_dif) == pytest.approx(cfg.log_std_max - cfg.log_std_min)
    assert torch.equal(wm2._action_masks, wm._action_masks)
    # a stale value inside an old file is overridden by the model's own, like the reference does
    old["log_std_min"] = torch.tensor(-3.0)
    wm3 = WorldModel(cfg)
    wm3.load_state_dict(old)
    assert float(wm3.log_std_min) == float(cfg.log_std_min)


def _rgb_cfg():
    cfg = named_config("c1")
    cfg.obs = "rgb"
    cfg.obs_shape = {"rgb": (9, 64, 64)}  # 3 stacked frames, reference envs/wrappers/pixels.py
    return cfg


def test_pixel_world_model_has_reference_keys_and_shapes():
    """(f)4: the conv encoder (reference layers.py:136-150) as a host-side module with the reference's state-dict keys."""
    cfg = _rgb_cfg()
    wm = WorldModel(cfg).eval()
    keys = set(wm.state_dict())
    for i, shp in ((2, (32, 9, 7, 7)), (4, (32, 32, 5, 5)), (6, (32, 32, 3, 3)), (8, (32, 32, 3, 3))):
        assert f"_encoder.rgb.{i}.weight" in keys and f"_encoder.rgb.{i}.bias" in keys
        assert tuple(wm.state_dict()[f"_encoder.rgb.{i}.weight"].shape) == shp
    obs = torch.randint(0, 256, (2, 9, 64, 64)).float()
    with torch.no_grad():
        z = wm.encode(obs, None)
        assert z.shape == (2, cfg.latent_dim)
        assert torch.allclose(z.view(2, -1, 8).sum(-1), torch.ones(2, 64), atol=1e-5)  # SimNorm output
        seq = wm.encode(obs.unsqueeze(0).repeat(3, 1, 1, 1, 1), None)  # [T, B, C, H, W] branch (world_model.py:110-111)
        assert seq.shape == (3, 2, cfg.latent_dim)


def test_pixel_modules_match_the_reference_modules():
    """ShiftAug / PixelPreprocess / conv against the reference's own modules on the same seed (build container only)."""
    from oracle import ref_runner

    if not ref_runner.available():
        pytest.skip("reference tree not present")
    from tdmpc2_amd import layers

    ref = ref_runner._import_reference().layers
    x = torch.randint(0, 256, (4, 9, 64, 64)).float()
    torch.manual_seed(11)
    want = ref.ShiftAug()(x)
    torch.manual_seed(11)
    got = layers.ShiftAug()(x)
    assert torch.equal(got, want)
    assert torch.equal(layers.PixelPreprocess()(x), ref.PixelPreprocess()(x))
    torch.manual_seed(0)
    mine = layers.conv((9, 64, 64), 32, act=layers.SimNorm(8))
    theirs = ref.conv((9, 64, 64), 32)
    theirs.load_state_dict(mine.state_dict())  # same module indices -> same keys
    torch.manual_seed(5)
    a = mine(x)
    torch.manual_seed(5)
    b = layers.SimNorm(8)(theirs(x))
    assert torch.allclose(a, b, atol=1e-6)


def test_planner_seed_differs_across_ranks(monkeypatch):
    """env-sharded ranks built from one cfg must not share the exploration-noise stream."""
    from tdmpc2_amd import tdmpc2 as t

    monkeypatch.setenv("RANK", "0")
    s0 = (t._rank() << 32) ^ 7
    monkeypatch.setenv("RANK", "3")
    s3 = (t._rank() << 32) ^ 7
    assert s0 != s3 and (s3 >> 32) == 3 and (s3 & 0xFFFFFFFF) == 7


def test_task_tables_of_the_agent_match_the_embedding_lookup():
    """TDMPC2._task_tables (what the training-side forwards index with `task`): rows equal what the model's own
    nn.Embedding(max_norm=1) returns at lookup (world_model.py:88-101), masks and discounts are the model's tables."""
    from tdmpc2_amd.config import named_config
    from tdmpc2_amd.tdmpc2 import TDMPC2

    cfg = named_config("small", task="mt30")
    agent = TDMPC2(cfg, device=torch.device("cpu"))
    with torch.no_grad():
        agent.model._task_emb.weight.mul_(3.0)  # rows of norm > 1: the renorm must bite
        agent.model._task_emb.weight[0].mul_(0.01)  # and one row that stays as it is
    w_before = agent.model._task_emb.weight.detach().clone()
    emb, mask, disc = agent._task_tables()
    ids = torch.arange(w_before.shape[0])
    looked_up = torch.nn.functional.embedding(ids, w_before.clone(), max_norm=1.0)
    assert torch.allclose(emb, looked_up, atol=1e-7)
    assert (emb.norm(dim=-1) <= 1.0 + 1e-5).all() and torch.equal(emb[0], w_before[0])
    assert torch.equal(agent.model._task_emb.weight, w_before)  # the table is derived, the parameter untouched
    assert torch.equal(mask, agent.model._action_masks.float()) and torch.equal(disc, agent.discount.float())


def test_every_tuning_key_of_the_header_has_a_binding():
    """include/tdmpc2_plan.h's enum tdmpc2_tuning <-> NativePlanner.set_* (rows per workgroup, in-launch refit, cluster path)."""
    import os
    import re

    from tdmpc2_amd import native

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "include", "tdmpc2_plan.h")).read()
    keys = dict((k, int(v)) for k, v in re.findall(r"(TDMPC2_TUNE_[A-Z_]+) = (\d+)", hdr))
    assert keys == {"TDMPC2_TUNE_ROWS_PER_WORKGROUP": 0, "TDMPC2_TUNE_FOLD_REFIT": 1, "TDMPC2_TUNE_CLUSTER": 2, "TDMPC2_TUNE_FUSE_LN": 3,
                    "TDMPC2_TUNE_REARM_AFTER": 4, "TDMPC2_TUNE_SAFE_ONCE": 5, "TDMPC2_TUNE_KSPLIT": 6, "TDMPC2_TUNE_FEWROW": 7,
                    "TDMPC2_TUNE_WAIT_US": 8, "TDMPC2_TUNE_EXPERT": 100}
    src = open(native.__file__).read()
    for key_id, method in [(1, "set_fold_refit"), (2, "set_cluster"), (3, "set_fuse_ln"), (4, "set_rearm_after"), (5, "plan_safely_once"), (6, "set_ksplit"),
                           (7, "set_fewrow"), (8, "set_wait_us")]:  # (TDMPC2_TUNE_EXPERT: set_expert, tests/test_abi.py)
        body = src[src.index(f"def {method}("):]
        body = body[:body.index("\n    def ", 10)]
        assert f"tdmpc2_plan_set_tuning(self._h, {key_id}," in body, method


@pytest.mark.parametrize("cfg_name,over", [("tiny", {}), ("tiny", dict(task="mt30")), ("c1", {})])
def test_checkpoint_conversion_agrees_with_the_references_own(cfg_name, over):
    """`checkpoint.convert_state_dict` + the WorldModel load hook against the reference's `api_model_conversion`
    (tdmpc2/common/layers.py:167-221), imported and run as it is (needs /root/reference: build container only).  An
    old-API checkpoint as released (flat-numbered Q keys, no log_std_* / _action_masks) goes through both; every key the
    reference produces must exist here with the same tensor, and no Q key may be left over."""
    from oracle import ref_runner

    if not ref_runner.available():
        pytest.skip("the reference tree is not on this machine")
    api_model_conversion = ref_runner._import_reference().layers.api_model_conversion
    cfg = named_config(cfg_name, **over)
    src = WorldModel(cfg)
    syn = {k: torch.as_tensor(v) for k, v in synth.make_state_dict(cfg, 0).items()}
    src.load_state_dict(dict(syn))
    old = checkpoint.to_old_format(src.state_dict())
    for k in ("log_std_min", "log_std_dif", "_action_masks"):
        old.pop(k, None)
    assert checkpoint.is_old_format(old)
    # reference: target_state_dict = the freshly built model's state dict (tdmpc2.py:92-94)
    fresh = WorldModel(cfg)
    want = api_model_conversion(fresh.state_dict(), dict(old))
    # here: strict load of the same old dict (the pre-hook converts), then read the model back
    mine = WorldModel(cfg)
    mine.load_state_dict(dict(old))
    got = mine.state_dict()
    assert set(want) == set(got), (sorted(set(want) - set(got)), sorted(set(got) - set(want)))
    for k, v in want.items():
        if torch.is_tensor(v):
            assert torch.equal(v, got[k]), k
        else:
            assert str(v) == str(got[k]) or k.endswith("__device"), (k, v, got[k])
    # ... and the function-level converter maps the Q keys exactly like the reference's renaming block
    conv = checkpoint.convert_state_dict(dict(old))
    for k, v in want.items():
        if "Qs" in k and torch.is_tensor(v) and not checkpoint.is_meta_key(k):
            assert torch.equal(conv[k], v), k
    # a new-format dict passes through both unchanged (layers.py:171-173)
    new = src.state_dict()
    assert api_model_conversion(fresh.state_dict(), dict(new)).keys() == new.keys()
    again = WorldModel(cfg)
    again.load_state_dict(dict(new))
    assert all(torch.equal(v, again.state_dict()[k]) for k, v in new.items() if torch.is_tensor(v))
