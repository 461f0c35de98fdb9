"""-m gpu: the packed weight file: tdmpc2_plan_export_packed / import_packed round trip --
a handle restored from the blob plans bit-identically to the handle that packed the checkpoint itself; mismatching
blobs are refused."""
import numpy as np
import pytest
import torch

from tests.gpu_common import case_on_gpu, dev, plan_inputs

pytestmark = pytest.mark.gpu


def _plan(c, model, planner):
    inp = plan_inputs(c, model)
    a = planner.plan(inp["z0"], inp["disc_pow"], inp["prev_mean"], inp["t0"], eval_mode=c["eval_mode"],
                     task_emb=inp["task_emb"], act_mask=inp["act_mask"], tape=inp["tape"])
    torch.cuda.synchronize()
    return a.cpu().numpy(), inp["prev_mean"].cpu().numpy()


@pytest.mark.parametrize("name,path,prec", [("c1", 1, 2), ("c1", 1, 1), ("mt5", 1, 2), ("small_ep", 2, 2), ("small_mt", 2, 1),
                                            ("c1_ep", 1, 2)])
def test_packed_round_trip_is_bit_identical(name, path, prec, tmp_path):
    from tdmpc2_amd.native import NativePlanner

    c, model, src = case_on_gpu(name, path, prec)
    want = _plan(c, model, src)
    fp = tmp_path / "planner.pack"
    src.save_packed(str(fp))
    dst = NativePlanner(c["cfg"], c["iterations"], dev(), max_envs=max(2, c["n_envs"]), path=path, precision=prec)
    dst.load_packed(str(fp))
    got = _plan(c, model, dst)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    # the target ensemble travelled too: td_target works on the restored handle without any bind
    if not c["cfg"].multitask:
        z = torch.as_tensor(c["z0"]).to(dev())
        r = torch.zeros(z.shape[0], device=dev())
        eps = torch.randn(z.shape[0], c["cfg"].action_dim, device=dev())
        qidx = torch.tensor([1, 0], dtype=torch.int32, device=dev())
        t0 = dst.td_target(z, r, r, 0.99, pi_eps=eps, qidx=qidx)
        t1 = src.td_target(z, r, r, 0.99, pi_eps=eps, qidx=qidx)
        assert torch.equal(t0, t1)
    dst.close()


def test_packed_with_encoder_and_run_obs():
    """The state encoder is part of the blob: run_obs on the restored handle equals run_obs on the source."""
    from oracle import cases
    from tdmpc2_amd import synth
    from tdmpc2_amd.native import NativePlanner
    from oracle import planner_oracle as po

    c = cases.build_case("c1")
    sd = {k: torch.as_tensor(v) for k, v in c["sd"].items()}
    model = po.OracleModel(c["cfg"], sd)
    src = NativePlanner(c["cfg"], c["iterations"], dev(), max_envs=c["n_envs"])
    src.bind_state_dict(sd)
    src.bind_encoder({k: v for k, v in sd.items() if k.startswith("_encoder.state.")})
    blob = src.export_packed()
    dst = NativePlanner(c["cfg"], c["iterations"], dev(), max_envs=c["n_envs"])
    dst.import_packed(blob, obs_dim=src.obs_dim)
    inp = plan_inputs(c, model)
    obs = torch.as_tensor(synth.make_obs(c["cfg"], c["n_envs"], seed=3)).to(dev())
    outs = []
    for pl in (src, dst):
        outs.append(pl.plan_obs(obs, inp["disc_pow"], inp["prev_mean"].clone(), inp["t0"], tape=inp["tape"]).cpu())
    assert torch.equal(outs[0], outs[1])
    src.close()
    dst.close()


def test_mismatching_blob_is_refused():
    from tdmpc2_amd.native import NativeError, NativePlanner

    c, model, src = case_on_gpu("c1", 1, 2)
    blob = src.export_packed()
    other = NativePlanner(c["cfg"], c["iterations"], dev(), max_envs=2, path=1, precision=1)  # other arithmetic
    with pytest.raises(NativeError, match="another model"):
        other.import_packed(blob)
    with pytest.raises(NativeError, match="magic|truncated"):
        other.import_packed(b"not a blob" * 100)
    with pytest.raises(NativeError):  # truncated data
        same = NativePlanner(c["cfg"], c["iterations"], dev(), max_envs=2, path=1, precision=2)
        same.import_packed(blob[: len(blob) // 2])
    unbound = NativePlanner(c["cfg"], c["iterations"], dev(), max_envs=2)
    with pytest.raises(NativeError, match="not bound"):
        unbound.export_packed()
