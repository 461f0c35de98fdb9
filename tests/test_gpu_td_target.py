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


def test_unsupported_handles():
    from tdmpc2_amd.native import NativeError

    c, model, planner = case_on_gpu("small")  # layered family
    z = torch.zeros(4, c["cfg"].latent_dim, device=dev())
    with pytest.raises(NativeError, match="fused"):
        planner.policy_value(z)
    c, model, planner = case_on_gpu("mt5")  # multitask
    z = torch.zeros(4, c["cfg"].latent_dim, device=dev())
    with pytest.raises(NativeError, match="single-task"):
        planner.policy_value(z)
