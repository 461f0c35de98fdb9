"""CPU: the C-ABI library builds, loads and exports every symbol include/tdmpc2_plan.h declares;
the host logic fails loudly without a GPU (no compute calls here)."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "tdmpc2_plan.h")


def _declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(tdmpc2_[a-z_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from tdmpc2_amd import native

    if not os.path.exists(native.lib_path()):
        subprocess.run([os.path.join(ROOT, "tdmpc2_amd", "csrc", "build.sh")], check=True)
    return ctypes.CDLL(native.lib_path())


def test_header_and_binding_agree():
    from tdmpc2_amd import native

    assert _declared_symbols() == sorted(native.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    from tdmpc2_amd import native

    for sym in _declared_symbols():
        assert hasattr(lib, sym), sym
    lib.tdmpc2_plan_abi_version.restype = ctypes.c_int
    assert lib.tdmpc2_plan_abi_version() == native.ABI_VERSION


def test_cfg_struct_matches_header_layout():
    from tdmpc2_amd import native

    # 12 int32 + 7 float + 7 int32, no padding
    assert ctypes.sizeof(native.PlanCfg) == 26 * 4
    assert ctypes.sizeof(native.Noise) == 6 * ctypes.sizeof(ctypes.c_void_p)
    assert ctypes.sizeof(native.Debug) == 6 * ctypes.sizeof(ctypes.c_void_p)


def test_invalid_arguments_report_errors(lib):
    lib.tdmpc2_plan_create.restype = ctypes.c_int
    lib.tdmpc2_last_error.restype = ctypes.c_char_p
    assert lib.tdmpc2_plan_create(None, None) != 0
    assert b"null" in lib.tdmpc2_last_error()


def test_no_cpu_fallback():
    import torch
    from tdmpc2_amd.config import named_config
    from tdmpc2_amd.native import NativeError, NativePlanner

    with pytest.raises(NativeError):
        NativePlanner(named_config("c1"), 6, torch.device("cpu"))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under tdmpc2_amd/ may reference it."""
    pkg = os.path.join(ROOT, "tdmpc2_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "/root/reference" not in txt, f


def test_new_entry_points_reject_null_arguments(lib):
    """No GPU needed: argument validation comes first in every entry point."""
    lib.tdmpc2_last_error.restype = ctypes.c_char_p
    for name, nargs in [("tdmpc2_plan_encode", 7), ("tdmpc2_plan_run_obs", 14), ("tdmpc2_plan_bind_encoder", 10),
                        ("tdmpc2_plan_policy_value", 11), ("tdmpc2_plan_td_target", 11)]:
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        if name == "tdmpc2_plan_td_target":
            fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float,
                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p]
            rc = fn(None, 1, None, None, None, 0.99, None, None, 0, None, None)
        else:
            rc = fn(*([None] + [0] * (nargs - 1))) if name != "tdmpc2_plan_run_obs" else fn(*([None] * nargs))
        assert rc == 1, name  # TDMPC2_ERR_INVALID
        assert b"null" in lib.tdmpc2_last_error()


def test_create_validates_the_configuration_before_touching_the_device(lib):
    """Configuration errors are reported with their own codes and messages (no GPU needed: validation comes first)."""
    from tdmpc2_amd import native
    from tdmpc2_amd.config import named_config

    lib.tdmpc2_plan_create.restype = ctypes.c_int
    lib.tdmpc2_plan_create.argtypes = [ctypes.POINTER(native.PlanCfg), ctypes.POINTER(ctypes.c_void_p)]
    lib.tdmpc2_last_error.restype = ctypes.c_char_p

    def make(**over):
        cfg = named_config("c1")
        f = dict(horizon=cfg.horizon, num_samples=cfg.num_samples, num_elites=cfg.num_elites, num_pi_trajs=cfg.num_pi_trajs,
                 iterations=6, action_dim=cfg.action_dim, latent_dim=cfg.latent_dim, mlp_dim=cfg.mlp_dim, task_dim=0,
                 num_bins=cfg.num_bins, num_q=cfg.num_q, simnorm_dim=8, vmin=-10.0, vmax=10.0, min_std=0.05, max_std=2.0,
                 temperature=0.5, log_std_min=-10.0, log_std_dif=12.0, multitask=0, episodic=0, max_envs=1, device=0, path=0,
                 precision=0)
        f.update(over)
        h = ctypes.c_void_p()
        rc = lib.tdmpc2_plan_create(ctypes.byref(native.PlanCfg(**f)), ctypes.byref(h))
        return rc, lib.tdmpc2_last_error().decode()

    INVALID, UNSUPPORTED = 1, 2
    rc, msg = make(path=1, latent_dim=768)  # fused family asked for a 768-wide model
    assert rc == UNSUPPORTED and "fused planner kernels" in msg
    rc, msg = make(path=2, num_samples=500)  # layered family: rows per plan must be a multiple of the GEMM tile
    assert rc == UNSUPPORTED and "num_samples" in msg
    rc, msg = make(multitask=1, task_dim=0)
    assert rc == INVALID and "task_dim" in msg
    rc, msg = make(multitask=1, task_dim=64, episodic=1)
    assert rc == UNSUPPORTED and "termination" in msg
    rc, msg = make(iterations=0)
    assert rc == INVALID
    rc, msg = make(precision=7)
    assert rc == INVALID and "precision" in msg
    rc, msg = make(path=9)
    assert rc == INVALID and "path" in msg


def test_the_shipped_library_reads_six_environment_variables():
    """the product library's environment surface is six names, listed in INTEGRATION.md section C; the test
    hooks live in libtdmpc2_plan_hooks.so only, the measurement knobs behind tdmpc2_plan_set_tuning (TDMPC2_TUNE_EXPERT)."""
    import re
    import subprocess

    from tdmpc2_amd import native

    if not os.path.exists(native.lib_path()):
        pytest.skip("library not built")
    names = lambda p: sorted(set(re.findall(r"^TDMPC2_[A-Z0-9_]+$", subprocess.run(["strings", p], capture_output=True, text=True).stdout, re.M)))  # noqa: E731
    shipped = names(native.lib_path())
    assert shipped == ["TDMPC2_CLUSTER", "TDMPC2_DEBUG_FAULT", "TDMPC2_FEWROW", "TDMPC2_FUSE_LN", "TDMPC2_KSPLIT", "TDMPC2_ONE_STREAM"], shipped
    doc = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    for n in shipped:
        assert f"`{n}" in doc, n
    if os.path.exists(native.hooks_lib_path()):
        assert set(names(native.hooks_lib_path())) - set(shipped) == set(native.TEST_HOOK_ENVS)
    # every expert knob of the header has a name in the binding, in order
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "tdmpc2_plan.h")).read()
    enum = hdr[hdr.index("enum tdmpc2_expert_knob"):]
    enum = enum[:enum.index("};")]
    assert tuple(re.findall(r"TDMPC2_X_([A-Z0-9_]+)", enum))[:-1] == native.EXPERT_KNOBS
