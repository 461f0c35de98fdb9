"""-m gpu: the training-side forward pieces on the planner's kernels: tdmpc2_plan_td_target against
the reference's own `TDMPC2._td_target` output (tests/golden/*.npz: `td_target`) and tdmpc2_plan_policy_value against
the oracle restatement of `update_pi`'s forward half."""
import numpy as np
import pytest
import torch

from tests.gpu_common import case_on_gpu, dev
from tests.helpers import load_golden

pytestmark = pytest.mark.gpu

PRECS = [1, 2]  # exact-fp32 MFMA, f16x2 split
TD_RTOL = 1e-4  # north star: within 1e-4 of the reference, relative to max(1, |v|)


def _batch(c):
    from oracle import cases

    tb = cases.td_batch(c["cfg"])
    d = dev()
    flat = lambda a: torch.as_tensor(a).reshape(-1, a.shape[-1]).to(d).contiguous()
    return tb, dict(next_z=flat(tb["next_z"]), reward=flat(tb["reward"])[:, 0].contiguous(),
                    terminated=flat(tb["terminated"])[:, 0].contiguous(), pi_eps=flat(tb["pi_eps"]),
                    qidx=torch.as_tensor(tb["qidx"]).to(d))


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", ["c1", "c2", "c1_wide"])
def test_td_target_matches_reference_golden(name, prec):
    c, model, planner = case_on_gpu(name, 1, prec)
    tb, b = _batch(c)
    want = load_golden(name)["td_target"].reshape(-1)
    td = planner.td_target(b["next_z"], b["reward"], b["terminated"], c["discounts"][0], b["pi_eps"], b["qidx"]).cpu().numpy()
    assert td.shape == want.shape  # 3 x 40 = 120 rows: one full 64-row workgroup and a ragged one
    err = np.abs(td - want) / np.maximum(1.0, np.abs(want))
    print(f"[{name} prec {prec}] td_target max rel err vs the reference's _td_target = {err.max():.2e}")
    assert err.max() < TD_RTOL


@pytest.mark.parametrize("prec", PRECS)
def test_policy_value_matches_oracle(prec):
    from oracle import planner_oracle as po

    c, model, planner = case_on_gpu("c2", 1, prec)
    tb, b = _batch(c)
    wa, wq = po.policy_value(model, torch.as_tensor(tb["next_z"]), None, torch.as_tensor(tb["pi_eps"]), torch.as_tensor(tb["qidx"]))
    a, q = planner.policy_value(b["next_z"], use_target=False, reduce="avg", pi_eps=b["pi_eps"], qidx=b["qidx"])
    ea = (a.cpu() - wa.reshape(-1, wa.shape[-1])).abs().max().item()
    eq = ((q.cpu() - wq.reshape(-1)).abs() / wq.reshape(-1).abs().clamp(min=1)).max().item()
    print(f"[c2 prec {prec}] policy_value: action max err {ea:.2e}, q max rel err {eq:.2e}")
    assert ea < 2e-5 and eq < TD_RTOL
    # the target ensemble is a different parameter set (synthetic weights draw it independently)
    _, qt = planner.policy_value(b["next_z"], use_target=True, reduce="avg", pi_eps=b["pi_eps"], qidx=b["qidx"], return_action=False)
    assert (qt - q).abs().max().item() > 1e-2
    # 'min' <= 'avg' row by row
    _, qm = planner.policy_value(b["next_z"], use_target=False, reduce="min", pi_eps=b["pi_eps"], qidx=b["qidx"], return_action=False)
    assert bool((qm <= q + 1e-6).all())


def test_sizes_and_in_library_noise():
    """1 row, 64, 65, 1000 rows; Philox noise and head draw inside the library."""
    c, model, planner = case_on_gpu("c1", 1, 2)
    from tdmpc2_amd import synth

    for rows in (1, 64, 65, 1000):
        z = torch.as_tensor(synth.make_latents(c["cfg"], rows, seed=rows)).to(dev())
        a, q = planner.policy_value(z, seed=rows)
        assert a.shape == (rows, c["cfg"].action_dim) and q.shape == (rows,)
        assert torch.isfinite(a).all() and torch.isfinite(q).all() and a.abs().max() <= 1
        td = planner.td_target(z, torch.zeros(rows, device=dev()), torch.ones(rows, device=dev()), 0.99, seed=3)
        assert torch.equal(td, torch.zeros_like(td))  # terminated rows: td = reward


def _mt_tables(c, model):
    """The per-task tables the reference indexes with `task` (world_model.py:88-101; tdmpc2.py:35-37)."""
    from tdmpc2_amd.config import get_discount

    cfg = c["cfg"]
    emb = model.sd["_task_emb.weight"]
    norm = emb.norm(2, dim=-1, keepdim=True)
    emb = torch.where(norm > 1.0, emb * (1.0 / (norm + 1e-7)), emb)  # nn.Embedding(max_norm=1)
    disc = torch.tensor([get_discount(cfg, ln) for ln in cfg.episode_lengths], dtype=torch.float32)
    return emb.to(dev()).contiguous(), model.sd["_action_masks"].to(dev()).contiguous(), disc.to(dev())


def _task_rows(tb, H):
    return torch.as_tensor(tb["tasks"]).repeat(H).to(torch.int32).to(dev()).contiguous()  # task [B] over next_z [H, B, L]


# (case, kernel family): every family / task mode combination against the reference's own `_td_target` output
TD_CASES = [("mt5", 1), ("mt5", 2), ("small", 2), ("small_ep", 2), ("small_ep_fire", 2), ("small_mt", 2), ("c1", 2), ("c2", 2), ("c3", 2), ("c4", 2),
            ("c1_ep", 1)]


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name,path", TD_CASES)
def test_td_target_multitask_and_layered_match_reference_golden(name, path, prec):
    """SURVEY 8(f) rank 2 beyond the single-task fused case: one task per row (multitask), the layered family, episodic
    handles -- against the reference-minted fixtures (oracle/make_golden.py: td_target)."""
    from tests.helpers import record_parity

    c, model, planner = case_on_gpu(name, path, prec)
    assert planner.path == path
    cfg = c["cfg"]
    tb, b = _batch(c)
    want = load_golden(name)["td_target"].reshape(-1)
    kw = {}
    if cfg.multitask:
        emb, mask, disc = _mt_tables(c, model)
        kw = dict(task_ids=_task_rows(tb, cfg.horizon), task_emb_table=emb, act_mask_table=mask)
    else:
        disc = c["discounts"][0]
    td = planner.td_target(b["next_z"], b["reward"], b["terminated"], disc, b["pi_eps"], b["qidx"], **kw).cpu().numpy()
    assert td.shape == want.shape
    err = np.abs(td - want) / np.maximum(1.0, np.abs(want))
    print(f"[{name} path {path} prec {prec}] td_target max rel err vs the reference's _td_target = {err.max():.2e}")
    record_parity(f"{name}/{'fused' if path == 1 else 'layered'}/{'fp32' if prec == 1 else 'split'}/td_target", value_rel=err.max())
    assert err.max() < TD_RTOL


@pytest.mark.parametrize("name,path", [("mt5", 1), ("small_mt", 2), ("small", 2)])
def test_policy_value_multitask_and_layered_match_oracle(name, path):
    from oracle import planner_oracle as po

    c, model, planner = case_on_gpu(name, path, 2)
    cfg = c["cfg"]
    tb, b = _batch(c)
    task = torch.as_tensor(tb["tasks"]).repeat(cfg.horizon) if cfg.multitask else None
    z2 = torch.as_tensor(tb["next_z"]).reshape(-1, cfg.latent_dim)
    e2 = torch.as_tensor(tb["pi_eps"]).reshape(-1, cfg.action_dim)
    wa, wq = po.policy_value(model, z2, task, e2, torch.as_tensor(tb["qidx"]))
    kw = {}
    if cfg.multitask:
        emb, mask, _ = _mt_tables(c, model)
        kw = dict(task_ids=_task_rows(tb, cfg.horizon), task_emb_table=emb, act_mask_table=mask)
    a, q = planner.policy_value(b["next_z"], use_target=False, reduce="avg", pi_eps=b["pi_eps"], qidx=b["qidx"], **kw)
    ea = (a.cpu() - wa).abs().max().item()
    eq = ((q.cpu() - wq.reshape(-1)).abs() / wq.reshape(-1).abs().clamp(min=1)).max().item()
    print(f"[{name} path {path}] policy_value: action max err {ea:.2e}, q max rel err {eq:.2e}")
    assert ea < 2e-5 and eq < TD_RTOL
    if cfg.multitask:  # masked action dimensions are exactly zero (world_model.py:163-167)
        m = mask[kw["task_ids"].long()]
        assert torch.equal(a * (1 - m), torch.zeros_like(a))


def test_value_argument_errors_are_loud():
    from tdmpc2_amd.native import NativeError

    c, model, planner = case_on_gpu("mt5", 1, 2)  # multitask handle without the task tables
    z = torch.zeros(4, c["cfg"].latent_dim, device=dev())
    with pytest.raises(ValueError, match="task_ids"):
        planner.policy_value(z)
    c, model, planner = case_on_gpu("c1", 1, 2)  # single-task handle with task ids
    with pytest.raises(ValueError, match="single-task"):
        planner.policy_value(z, task_ids=torch.zeros(4, dtype=torch.int32, device=dev()))
    c, model, planner = case_on_gpu("small", 2, 2)  # layered workspace: at most max_envs * num_samples rows
    zz = torch.zeros(2 * 128 + 129, c["cfg"].latent_dim, device=dev())
    with pytest.raises(NativeError, match="workspace"):
        planner.policy_value(zz)
