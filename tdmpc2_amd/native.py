"""ctypes binding of the planner's C ABI (include/tdmpc2_plan.h).

The shared library `tdmpc2_amd/libtdmpc2_plan.so` is built in-tree by
`tdmpc2_amd/csrc/build.sh` (hipcc, gfx950).  There is no CPU fallback: if the
library is missing, or a planner is requested on a non-GPU device, this module
raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch

# TDMPC2_PLAN_LIB points profiling runs at an ablation build of the same library (tools/ablate.sh)
_LIB_PATH = os.environ.get("TDMPC2_PLAN_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtdmpc2_plan.so")
_lib = None

ABI_VERSION = 9

# every symbol include/tdmpc2_plan.h declares (tests check the .so exports all of them)
ABI_SYMBOLS = [
    "tdmpc2_plan_abi_version", "tdmpc2_last_error", "tdmpc2_plan_create", "tdmpc2_plan_destroy",
    "tdmpc2_plan_device_bytes", "tdmpc2_plan_path", "tdmpc2_plan_precision", "tdmpc2_plan_bind_weights", "tdmpc2_plan_run", "tdmpc2_plan_estimate_value",
    "tdmpc2_plan_estimate_value_trace", "tdmpc2_plan_refit", "tdmpc2_plan_set_tuning", "tdmpc2_plan_set_profiling",
    "tdmpc2_plan_profile_read", "tdmpc2_plan_bind_encoder", "tdmpc2_plan_encode", "tdmpc2_plan_run_obs",
    "tdmpc2_plan_policy_value", "tdmpc2_plan_td_target", "tdmpc2_plan_policy_value_mt", "tdmpc2_plan_td_target_mt",
    "tdmpc2_plan_packed_size", "tdmpc2_plan_export_packed", "tdmpc2_plan_import_packed",
    "tdmpc2_plan_shard_begin", "tdmpc2_plan_shard_values", "tdmpc2_plan_shard_refit",
    "tdmpc2_plan_export_noise", "tdmpc2_plan_call_counter", "tdmpc2_plan_set_call_counter", "tdmpc2_plan_take_fault",
    "tdmpc2_plan_fault_info", "tdmpc2_plan_fault_word",
]

NET_DYNAMICS, NET_REWARD, NET_PI, NET_Q, NET_TERMINATION, NET_TARGET_Q = range(6)
PATH_AUTO, PATH_FUSED, PATH_LAYERED = range(3)  # enum tdmpc2_path
PREC_AUTO, PREC_FP32, PREC_SPLIT_F16 = range(3)  # enum tdmpc2_precision


class FaultInfo(C.Structure):  # struct tdmpc2_fault_info
    _fields_ = [("faults_total", C.c_int32), ("rearms", C.c_int32), ("degraded", C.c_int32), ("clean_calls", C.c_int32),
                ("rearm_after", C.c_int32), ("reserved", C.c_int32), ("seconds_since_fault", C.c_double)]


class PlanCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("horizon", "num_samples", "num_elites", "num_pi_trajs", "iterations",
                                          "action_dim", "latent_dim", "mlp_dim", "task_dim", "num_bins", "num_q",
                                          "simnorm_dim")] + \
               [(n, C.c_float) for n in ("vmin", "vmax", "min_std", "max_std", "temperature", "log_std_min",
                                         "log_std_dif")] + \
               [(n, C.c_int32) for n in ("multitask", "episodic", "max_envs", "device", "path", "precision", "num_valid_samples")]


class Noise(C.Structure):
    _fields_ = [("pi_traj_eps", C.c_void_p), ("sample_eps", C.c_void_p), ("pi_eps", C.c_void_p),
                ("qidx", C.c_void_p), ("gumbel_exp", C.c_void_p), ("final_eps", C.c_void_p)]


class TaskTables(C.Structure):
    """struct tdmpc2_task_tables: one task per row of a training batch + the per-task tables."""
    _fields_ = [("task_ids", C.c_void_p), ("task_emb", C.c_void_p), ("act_mask", C.c_void_p), ("discount", C.c_void_p),
                ("n_tasks", C.c_int32)]


class Debug(C.Structure):
    _fields_ = [("value", C.c_void_p), ("elite_idx", C.c_void_p), ("score", C.c_void_p), ("mean", C.c_void_p),
                ("std", C.c_void_p), ("actions", C.c_void_p)]


class NativeError(RuntimeError):
    pass


def lib_path() -> str:
    return _LIB_PATH


# Environment variables that are TEST HOOKS of the bounded-wait / stream-order machinery.  The shipped library does not read them
# (ABI 9); `libtdmpc2_plan_hooks.so` -- the same objects with the C-ABI unit compiled -DTDMPC2_TEST_HOOKS, built beside the product
# library -- does, and a planner created while one of them is set is created on that library (the GPU tests of the fault paths).
TEST_HOOK_ENVS = ("TDMPC2_CLUSTER_FAULT", "TDMPC2_DEBUG_NO_TURN", "TDMPC2_POISON")
_HOOKS_LIB_PATH = os.path.join(os.path.dirname(_LIB_PATH), "libtdmpc2_plan_hooks.so")
_lib_hooks = None


def hooks_lib_path() -> str:
    return _HOOKS_LIB_PATH


def load_library(hooks: bool = False):
    """dlopen the planner library (hooks = True: its test-hooks flavour) and declare its prototypes.  Raises if absent."""
    global _lib, _lib_hooks
    if hooks and "TDMPC2_PLAN_LIB" not in os.environ:
        if _lib_hooks is None:
            _lib_hooks = _open(_HOOKS_LIB_PATH)
        return _lib_hooks
    if _lib is None:
        _lib = _open(_LIB_PATH)
    return _lib


def _open(path):
    if not os.path.exists(path):
        raise NativeError(f"{path} not found: build it with tdmpc2_amd/csrc/build.sh "
                          "(or __graft_entry__.build()); there is no CPU fallback for the planner")
    lib = C.CDLL(path)
    vp, i32, u64 = C.c_void_p, C.c_int, C.c_uint64
    lib.tdmpc2_plan_abi_version.restype = i32
    lib.tdmpc2_last_error.restype = C.c_char_p
    lib.tdmpc2_plan_create.argtypes = [C.POINTER(PlanCfg), C.POINTER(vp)]
    lib.tdmpc2_plan_create.restype = i32
    lib.tdmpc2_plan_destroy.argtypes = [vp]
    lib.tdmpc2_plan_destroy.restype = None
    lib.tdmpc2_plan_device_bytes.argtypes = [vp]
    lib.tdmpc2_plan_device_bytes.restype = u64
    lib.tdmpc2_plan_path.argtypes = [vp]
    lib.tdmpc2_plan_path.restype = i32
    lib.tdmpc2_plan_precision.argtypes = [vp]
    lib.tdmpc2_plan_precision.restype = i32
    lib.tdmpc2_plan_bind_weights.argtypes = [vp, i32, i32, vp, vp, vp, vp, i32, i32, vp]
    lib.tdmpc2_plan_bind_weights.restype = i32
    lib.tdmpc2_plan_run.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, i32, C.POINTER(Noise), u64, vp,
                                    C.POINTER(Debug), vp]
    lib.tdmpc2_plan_run.restype = i32
    lib.tdmpc2_plan_bind_encoder.argtypes = [vp, i32, i32, vp, vp, vp, vp, i32, i32, vp]
    lib.tdmpc2_plan_bind_encoder.restype = i32
    lib.tdmpc2_plan_encode.argtypes = [vp, i32, vp, i32, vp, vp, vp]
    lib.tdmpc2_plan_encode.restype = i32
    lib.tdmpc2_plan_run_obs.argtypes = [vp, i32, vp, i32, vp, vp, vp, vp, vp, i32, C.POINTER(Noise), u64, vp, vp]
    lib.tdmpc2_plan_run_obs.restype = i32
    lib.tdmpc2_plan_policy_value.argtypes = [vp, i32, vp, i32, i32, vp, vp, u64, vp, vp, vp]
    lib.tdmpc2_plan_policy_value.restype = i32
    lib.tdmpc2_plan_td_target.argtypes = [vp, i32, vp, vp, vp, C.c_float, vp, vp, u64, vp, vp]
    lib.tdmpc2_plan_td_target.restype = i32
    lib.tdmpc2_plan_policy_value_mt.argtypes = [vp, i32, vp, C.POINTER(TaskTables), i32, i32, vp, vp, u64, vp, vp, vp]
    lib.tdmpc2_plan_policy_value_mt.restype = i32
    lib.tdmpc2_plan_td_target_mt.argtypes = [vp, i32, vp, vp, vp, C.c_float, C.POINTER(TaskTables), vp, vp, u64, vp, vp]
    lib.tdmpc2_plan_td_target_mt.restype = i32
    lib.tdmpc2_plan_packed_size.argtypes = [vp, C.POINTER(u64)]
    lib.tdmpc2_plan_packed_size.restype = i32
    lib.tdmpc2_plan_export_packed.argtypes = [vp, vp, u64, vp]
    lib.tdmpc2_plan_export_packed.restype = i32
    lib.tdmpc2_plan_import_packed.argtypes = [vp, vp, u64, vp]
    lib.tdmpc2_plan_import_packed.restype = i32
    lib.tdmpc2_plan_shard_begin.argtypes = [vp, i32, vp, vp, vp, vp, vp, C.POINTER(Noise), u64, vp]
    lib.tdmpc2_plan_shard_begin.restype = i32
    lib.tdmpc2_plan_shard_values.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, C.POINTER(Noise), u64, vp, vp]
    lib.tdmpc2_plan_shard_values.restype = i32
    lib.tdmpc2_plan_shard_refit.argtypes = [vp, i32, i32, vp, vp, vp, i32, C.POINTER(Noise), u64, vp, C.POINTER(Debug), vp]
    lib.tdmpc2_plan_shard_refit.restype = i32
    lib.tdmpc2_plan_estimate_value.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tdmpc2_plan_estimate_value.restype = i32
    lib.tdmpc2_plan_estimate_value_trace.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tdmpc2_plan_estimate_value_trace.restype = i32
    lib.tdmpc2_plan_refit.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.tdmpc2_plan_refit.restype = i32
    lib.tdmpc2_plan_export_noise.argtypes = [vp, i32, i32, u64, C.c_uint32, C.POINTER(Noise), vp]
    lib.tdmpc2_plan_export_noise.restype = i32
    lib.tdmpc2_plan_call_counter.argtypes = [vp, C.POINTER(C.c_uint32)]
    lib.tdmpc2_plan_call_counter.restype = i32
    lib.tdmpc2_plan_set_call_counter.argtypes = [vp, C.c_uint32]
    lib.tdmpc2_plan_set_call_counter.restype = i32
    lib.tdmpc2_plan_take_fault.argtypes = [vp, C.POINTER(i32)]
    lib.tdmpc2_plan_take_fault.restype = i32
    lib.tdmpc2_plan_fault_word.argtypes = [vp, vp, vp]
    lib.tdmpc2_plan_fault_word.restype = i32
    lib.tdmpc2_plan_fault_info.argtypes = [vp, C.POINTER(FaultInfo)]
    lib.tdmpc2_plan_fault_info.restype = i32
    lib.tdmpc2_plan_set_tuning.argtypes = [vp, i32, i32]
    lib.tdmpc2_plan_set_tuning.restype = i32
    lib.tdmpc2_plan_set_profiling.argtypes = [vp, i32]
    lib.tdmpc2_plan_set_profiling.restype = i32
    lib.tdmpc2_plan_profile_read.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_int)]
    lib.tdmpc2_plan_profile_read.restype = i32
    if lib.tdmpc2_plan_abi_version() != ABI_VERSION:
        raise NativeError(f"ABI version mismatch: library {lib.tdmpc2_plan_abi_version()}, binding {ABI_VERSION}")
    return lib


# include/tdmpc2_plan.h: enum tdmpc2_expert_knob, in order.  TDMPC2_X_<NAME> in the environment OF THE PYTHON PROCESS is applied to every
# planner at creation (the A/B tools: tools/gpu_env_ab.sh); the library itself reads none of these.
EXPERT_KNOBS = ("GEMM_W256_MIN", "GEMM_W_SPLIT_MIN", "GEMM_W_SPLIT_MAX", "GEMM_W_SPLIT_OVH", "KSPLIT_AUTO_LO", "KSPLIT_AUTO_MIN",
                "GEMM_W_XCD_ROWS", "GEMM_NCT1", "GEMM_WIDE_MIN", "GEMM_RT4", "GEMM_FILL_PERMILLE", "GEMM_FILL_HEAD_PERMILLE", "GEMM_SD1",
                "GEMM_XCD_ROWS", "GEMM_COL_PAD", "TWOHOT_UNFUSED", "Z0_SHARED_OFF", "MID_PARTS_MAX", "MID_FUSE_LN", "MID_SPLIT_XCD", "MID_PIFOLD")
TUNE_EXPERT = 100


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _chk_tensor(name, t, dtype, shape, device):
    if t.device != device:
        raise ValueError(f"{name}: expected device {device}, got {t.device}")
    if t.dtype != dtype:
        raise ValueError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: must be contiguous")
    return t


class NativePlanner:
    """Owns one `tdmpc2_plan_t` handle on one GPU.

    Mirrors what the reference keeps as planner state in `TDMPC2.__init__`
    (tdmpc2/tdmpc2.py:17-43): the world-model weights (re-packed on the device)
    and the workspace for `max_envs` concurrent plans.
    """

    def __init__(self, cfg, iterations: int, device: torch.device, max_envs: int = 1,
                 log_std_min: Optional[float] = None, log_std_dif: Optional[float] = None, path: int = PATH_AUTO,
                 precision: int = PREC_AUTO):
        device = torch.device(device)
        if device.type != "cuda":
            raise NativeError(f"the planner runs on an MI355X only (device {device}); there is no CPU fallback")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.lib = load_library(hooks=any(k in os.environ for k in TEST_HOOK_ENVS))
        self.cfg = cfg
        self.device = device
        self.iterations = int(iterations)
        self.max_envs = int(max_envs)
        lsmin = float(cfg.log_std_min) if log_std_min is None else float(log_std_min)
        lsdif = float(cfg.log_std_max) - float(cfg.log_std_min) if log_std_dif is None else float(log_std_dif)
        # cfg.num_samples may be anything (config.yaml:36); the kernels own 64- / 128-row tiles.  The handle is created with the
        # count rounded UP to 128 and told the true one (tdmpc2_plan_cfg::num_valid_samples): the padding rows are rolled out but
        # can never be elites.  self.cfg keeps the caller's count; tapes are padded and stages sliced at this boundary.
        n_true = int(cfg.num_samples)
        fused_ok = int(cfg.latent_dim) == 512 and int(cfg.mlp_dim) == 512 and int(path) != PATH_LAYERED
        tile = 64 if fused_ok else 128  # rows a workgroup owns: fused family 64, layered family 128
        self._npad = (n_true + tile - 1) // tile * tile
        c = PlanCfg(horizon=cfg.horizon, num_samples=self._npad, num_valid_samples=(n_true if self._npad != n_true else 0),
                    num_elites=cfg.num_elites,
                    num_pi_trajs=cfg.num_pi_trajs, iterations=self.iterations, action_dim=cfg.action_dim,
                    latent_dim=cfg.latent_dim, mlp_dim=cfg.mlp_dim, task_dim=cfg.task_dim, num_bins=cfg.num_bins,
                    num_q=cfg.num_q, simnorm_dim=cfg.simnorm_dim, vmin=cfg.vmin, vmax=cfg.vmax, min_std=cfg.min_std,
                    max_std=cfg.max_std, temperature=cfg.temperature, log_std_min=lsmin, log_std_dif=lsdif,
                    multitask=int(bool(cfg.multitask)), episodic=int(bool(cfg.episodic)), max_envs=self.max_envs,
                    device=device.index, path=int(path), precision=int(precision))
        h = C.c_void_p()
        with torch.cuda.device(device):  # the library restores the caller's device itself; this keeps torch's view in step
            self._check(self.lib.tdmpc2_plan_create(C.byref(c), C.byref(h)))
        self._h = h
        self.path = int(self.lib.tdmpc2_plan_path(h))  # PATH_FUSED or PATH_LAYERED
        self.precision = int(self.lib.tdmpc2_plan_precision(h))  # PREC_FP32 or PREC_SPLIT_F16
        for name in EXPERT_KNOBS:  # measurement knobs from THIS process's environment (see EXPERT_KNOBS)
            if f"TDMPC2_X_{name}" in os.environ:
                self.set_expert(name, int(os.environ[f"TDMPC2_X_{name}"]))
        self._seed_calls = 0
        self._shard_noise = None  # struct tdmpc2_noise of the sharded plan in progress (shard_begin .. shard_refit)
        self.encoder_layers = 0
        self.obs_dim = None

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc: int):
        if rc != 0:
            raise NativeError(f"tdmpc2_plan error {rc}: {self.lib.tdmpc2_last_error().decode()}")

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.tdmpc2_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def device_bytes(self) -> int:
        return int(self.lib.tdmpc2_plan_device_bytes(self._h))

    # ------------------------------------------------------------------ weights
    def bind_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Bind planner weights from a state dict in the reference's (new-format)
        checkpoint key layout: `_dynamics.{i}.*`, `_reward.{i}.*`, `_pi.{i}.*`,
        `_Qs.params.{i}.*` (stacked over num_q).  tdmpc2/common/layers.py:167-199."""
        nets = [(NET_DYNAMICS, "_dynamics"), (NET_REWARD, "_reward"), (NET_PI, "_pi"), (NET_Q, "_Qs.params")]
        if self.cfg.episodic:
            nets.append((NET_TERMINATION, "_termination"))
        if "_target_Qs_params.0.weight" in sd:
            nets.append((NET_TARGET_Q, "_target_Qs_params"))  # optional: td_target (tdmpc2.py:239-254)
        keep = []
        with torch.cuda.device(self.device):
            for net, prefix in nets:
                for layer in range(3):
                    def get(name, required=True):
                        k = f"{prefix}.{layer}.{name}"
                        if k not in sd:
                            if required:
                                raise KeyError(f"state dict lacks {k}")
                            return None
                        t = sd[k].detach().to(self.device, torch.float32).contiguous()
                        keep.append(t)
                        return t
                    W, b = get("weight"), get("bias")
                    g, beta = get("ln.weight", False), get("ln.bias", False)
                    out_f, in_f = int(W.shape[-2]), int(W.shape[-1])
                    self._check(self.lib.tdmpc2_plan_bind_weights(self._h, net, layer, _ptr(W), _ptr(b), _ptr(g),
                                                                  _ptr(beta), out_f, in_f, self._stream()))
            torch.cuda.current_stream(self.device).synchronize()  # sources may now be freed

    def bind_encoder(self, sd: Dict[str, torch.Tensor], prefix: str = "_encoder.state"):
        """Bind the state encoder (tdmpc2/common/layers.py:153-164) from checkpoint keys
        `_encoder.state.{i}.{weight,bias,ln.weight,ln.bias}`; afterwards `encode` / `plan_obs` run it in HIP."""
        n = 0
        while f"{prefix}.{n}.weight" in sd:
            n += 1
        if n == 0:
            raise KeyError(f"state dict has no {prefix}.0.weight")
        keep = []
        with torch.cuda.device(self.device):
            for layer in range(n):
                ts = []
                for name in ("weight", "bias", "ln.weight", "ln.bias"):
                    t = sd[f"{prefix}.{layer}.{name}"].detach().to(self.device, torch.float32).contiguous()
                    keep.append(t)
                    ts.append(t)
                W = ts[0]
                self._check(self.lib.tdmpc2_plan_bind_encoder(self._h, layer, n, _ptr(W), _ptr(ts[1]), _ptr(ts[2]), _ptr(ts[3]),
                                                              int(W.shape[0]), int(W.shape[1]), self._stream()))
            torch.cuda.current_stream(self.device).synchronize()
        self.encoder_layers = n
        self.obs_dim = int(sd[f"{prefix}.0.weight"].shape[1]) - int(self.cfg.task_dim)

    def encode(self, obs, task_emb=None, out: Optional[torch.Tensor] = None):
        """WorldModel.encode for state observations (world_model.py:103-112): obs [E, obs_dim] -> z [E, L]."""
        cfg, dev = self.cfg, self.device
        E = int(obs.shape[0])
        _chk_tensor("obs", obs, torch.float32, (E, self.obs_dim), dev)
        if cfg.multitask:
            if task_emb is None:
                raise ValueError("multitask encoding needs task_emb")
            _chk_tensor("task_emb", task_emb, torch.float32, (E, cfg.task_dim), dev)
        else:
            task_emb = None
        z = out if out is not None else torch.empty(E, cfg.latent_dim, device=dev, dtype=torch.float32)
        _chk_tensor("z", z, torch.float32, (E, cfg.latent_dim), dev)
        with torch.cuda.device(dev):
            self._check(self.lib.tdmpc2_plan_encode(self._h, E, _ptr(obs), self.obs_dim, _ptr(task_emb), _ptr(z), self._stream()))
        return z

    def plan_obs(self, obs, disc_pow, prev_mean, t0, eval_mode=False, task_emb=None, act_mask=None,
                 tape: Optional[Dict[str, torch.Tensor]] = None, seed: int = 0, out: Optional[torch.Tensor] = None):
        """TDMPC2._plan from the observation on (tdmpc2.py:152-206): encode + plan in one library call."""
        cfg, dev = self.cfg, self.device
        E = int(obs.shape[0])
        H, N, K, P, A, I = cfg.horizon, cfg.num_samples, cfg.num_elites, cfg.num_pi_trajs, cfg.action_dim, self.iterations
        _chk_tensor("obs", obs, torch.float32, (E, self.obs_dim), dev)
        _chk_tensor("disc_pow", disc_pow, torch.float32, (E, H + 1), dev)
        if cfg.multitask:
            if task_emb is None or act_mask is None:
                raise ValueError("multitask planning needs task_emb and act_mask")
            _chk_tensor("task_emb", task_emb, torch.float32, (E, cfg.task_dim), dev)
            _chk_tensor("act_mask", act_mask, torch.float32, (E, A), dev)
        _chk_tensor("prev_mean", prev_mean, torch.float32, (E, H, A), dev)
        _chk_tensor("t0", t0, torch.uint8, (E,), dev)
        action = out if out is not None else torch.empty(E, A, device=dev, dtype=torch.float32)
        _chk_tensor("action", action, torch.float32, (E, A), dev)
        noise_p = None
        if tape is not None:
            noise = self._noise(tape, E)
            noise_p = C.byref(noise)
        with torch.cuda.device(dev):
            self._check(self.lib.tdmpc2_plan_run_obs(self._h, E, _ptr(obs), self.obs_dim, _ptr(task_emb), _ptr(act_mask),
                                                     _ptr(disc_pow), _ptr(prev_mean), _ptr(t0), int(bool(eval_mode)), noise_p,
                                                     C.c_uint64(int(seed) & (2**64 - 1)), _ptr(action), self._stream()))
        return action

    def _noise(self, tape, E):
        shapes = self.noise_shapes(E)
        for k, (shp, dt) in shapes.items():
            _chk_tensor(f"tape[{k}]", tape[k], dt, shp, self.device)
        if self._npad != self.cfg.num_samples:  # pad the sample axis with zeros: the padding rows' draws are irrelevant
            cfg, N, NP, P = self.cfg, self.cfg.num_samples, self._npad, self.cfg.num_pi_trajs
            se = torch.zeros(E, self.iterations, cfg.horizon, NP - P, cfg.action_dim, device=self.device)
            se[:, :, :, :N - P] = tape["sample_eps"]
            pe = torch.zeros(E, self.iterations, NP, cfg.action_dim, device=self.device)
            pe[:, :, :N] = tape["pi_eps"]
            tape = dict(tape, sample_eps=se, pi_eps=pe)
            self._padded_tape = tape  # (kept alive until the next call)
        return Noise(**{k: tape[k].data_ptr() for k in shapes})

    def _whole_tiles_only(self, what):
        if self._npad != self.cfg.num_samples:
            raise NativeError(f"{what} is a stage-wise entry point: it needs num_samples ({self.cfg.num_samples}) to be a multiple of the "
                              f"kernels' row tile; plan() / plan_obs() pad to {self._npad} themselves")

    def noise_shapes(self, E):
        """Shapes / dtypes of the six tensors of a noise tape for E environments (struct tdmpc2_noise)."""
        cfg = self.cfg
        H, N, K, P, A, I = cfg.horizon, cfg.num_samples, cfg.num_elites, cfg.num_pi_trajs, cfg.action_dim, self.iterations
        return {"pi_traj_eps": ((E, H, P, A), torch.float32), "sample_eps": ((E, I, H, N - P, A), torch.float32),
                "pi_eps": ((E, I, N, A), torch.float32), "qidx": ((E, I, 2), torch.int32),
                "gumbel_exp": ((E, K), torch.float32), "final_eps": ((E, A), torch.float32)}

    def call_counter(self) -> int:
        """The handle's call counter: the value the NEXT plan / td_target / policy_value call mixes into its Philox key."""
        n = C.c_uint32()
        self._check(self.lib.tdmpc2_plan_call_counter(self._h, C.byref(n)))
        return int(n.value)

    def set_call_counter(self, value: int):
        self._check(self.lib.tdmpc2_plan_set_call_counter(self._h, C.c_uint32(int(value) & 0xFFFFFFFF)))

    def export_noise(self, seed: int, call: int, n_envs: int, env_first: int = 0, fields=None):
        """The draws a tape = None plan makes under (seed, call = call_counter() read BEFORE that plan) for environments
        [env_first, env_first + n_envs), as a noise-tape dict: feeding it back as `tape` reproduces the plan bit for bit,
        and the same tensors replay through the CPU oracle (tdmpc2_plan_export_noise)."""
        self._whole_tiles_only("export_noise")
        shapes = self.noise_shapes(n_envs)
        out = {k: torch.empty(shp, dtype=dt, device=self.device) for k, (shp, dt) in shapes.items() if fields is None or k in fields}
        noise = Noise(**{k: v.data_ptr() for k, v in out.items()})
        with torch.cuda.device(self.device):
            self._check(self.lib.tdmpc2_plan_export_noise(self._h, int(env_first), int(n_envs), C.c_uint64(int(seed) & (2**64 - 1)),
                                                          C.c_uint32(int(call) & 0xFFFFFFFF), C.byref(noise), self._stream()))
        return out

    def take_fault(self) -> int:
        """Number of calls invalidated by a bounded inter-workgroup wait that gave up (cluster path, fused NormedLinear epilogue)
        since the last take_fault; such a plan returned NaN actions and kept its prev_mean, such a td_target / policy_value
        returned NaN.  The handle runs the paths without waits until it re-arms (fault_info, set_rearm_after).  Call after a sync."""
        n = C.c_int()
        self._check(self.lib.tdmpc2_plan_take_fault(self._h, C.byref(n)))
        return int(n.value)

    def fault_word(self, dst):
        """tdmpc2_plan_fault_word: the device-visible verdict word of the calls in flight copied into dst[0] (int32, on this device)
        in stream order -- no host synchronisation (dist.sharded_plan appends it to the slice it all-gathers)."""
        _chk_tensor("dst", dst, torch.int32, (1,), self.device)
        with torch.cuda.device(self.device):
            self._check(self.lib.tdmpc2_plan_fault_word(self._h, _ptr(dst), self._stream()))

    def fault_info(self) -> dict:
        """The handle's fault history (tdmpc2_plan_fault_info): faults_total, rearms, degraded, clean_calls, rearm_after,
        seconds_since_fault (-1: never).  Nothing is consumed."""
        fi = FaultInfo()
        self._check(self.lib.tdmpc2_plan_fault_info(self._h, C.byref(fi)))
        return {k: getattr(fi, k) for k, _ in FaultInfo._fields_ if k != "reserved"}

    def set_expert(self, name: str, value: Optional[int]):
        """A measurement knob of the layered family's tile choice (TDMPC2_TUNE_EXPERT + tdmpc2_expert_knob); None = the default."""
        v = -2**31 if value is None else int(value)
        self._check(self.lib.tdmpc2_plan_set_tuning(self._h, TUNE_EXPERT + EXPERT_KNOBS.index(name), v))

    def set_fewrow(self, on):
        """Layered family (TDMPC2_TUNE_FEWROW): K-part tiles + row kernels for calls with few sample rows (single plans)."""
        self._check(self.lib.tdmpc2_plan_set_tuning(self._h, 7, int(bool(on))))

    def set_wait_us(self, us):
        """Wall-clock bound of the inter-workgroup waits in microseconds (TDMPC2_TUNE_WAIT_US; default 5000)."""
        self._check(self.lib.tdmpc2_plan_set_tuning(self._h, 8, int(us)))

    def set_ksplit(self, mode):
        """TDMPC2_TUNE_KSPLIT: layered family -- split 256 x 256 GEMM tiles along K over 2-4 workgroups: 0 never (a plan's bits do
        not depend on the size of the call it is part of), 1 whenever the round arithmetic says so, 2 (default) only for
        launches that leave most of the chip idle (single plans of the 317M model)."""
        self._check(self.lib.tdmpc2_plan_set_tuning(self._h, 6, int(mode)))

    def plan_safely_once(self, on: bool = True):
        """TDMPC2_TUNE_SAFE_ONCE: the next whole plan (plan(), or shard_begin .. the last shard_refit) runs on the paths without
        inter-workgroup waits; the settings asked for, the downgrade state and the re-arm counter stay as they are."""
        self._check(self.lib.tdmpc2_plan_set_tuning(self._h, 5, int(bool(on))))

    def set_rearm_after(self, clean_calls: int):
        """TDMPC2_TUNE_REARM_AFTER: clean calls after which a handle downgraded by a reported wait returns to the fast paths (0: never)."""
        self._check(self.lib.tdmpc2_plan_set_tuning(self._h, 4, int(clean_calls)))

    # ------------------------------------------------------------------ training-side forward pieces
    def _task_tables(self, R, task_ids, task_emb_table, act_mask_table, discount_table=None):
        """struct tdmpc2_task_tables for a multitask batch (None for single-task handles).  Returns (ctypes pointer | None,
        keep-alive tuple)."""
        cfg, dev = self.cfg, self.device
        if not cfg.multitask:
            if task_ids is not None:
                raise ValueError("task_ids given to a single-task planner")
            return None, ()
        if task_ids is None or task_emb_table is None or act_mask_table is None:
            raise ValueError("multitask policy_value / td_target need task_ids, task_emb_table and act_mask_table")
        n_tasks = int(task_emb_table.shape[0])
        _chk_tensor("task_ids", task_ids, torch.int32, (R,), dev)
        _chk_tensor("task_emb_table", task_emb_table, torch.float32, (n_tasks, cfg.task_dim), dev)
        _chk_tensor("act_mask_table", act_mask_table, torch.float32, (n_tasks, cfg.action_dim), dev)
        if discount_table is not None:
            _chk_tensor("discount_table", discount_table, torch.float32, (n_tasks,), dev)
        tt = TaskTables(task_ids=task_ids.data_ptr(), task_emb=task_emb_table.data_ptr(), act_mask=act_mask_table.data_ptr(),
                        discount=None if discount_table is None else discount_table.data_ptr(), n_tasks=n_tasks)
        return C.byref(tt), (tt, task_ids, task_emb_table, act_mask_table, discount_table)

    def policy_value(self, z, use_target=False, reduce="avg", pi_eps=None, qidx=None, seed: int = 0, return_action=True,
                     task_ids=None, task_emb_table=None, act_mask_table=None):
        """a = pi(z), then two Q heads of the online / target ensemble, 'avg' or 'min' (the forward half of
        TDMPC2.update_pi, tdmpc2.py:208-225).  z [R, L] -> (action [R, A] or None, q [R]).  Multitask models: one task
        per row (`task_ids` int32 [R]) + the model's embedding / action-mask tables (world_model.py:88-101)."""
        cfg, dev = self.cfg, self.device
        R = int(z.shape[0])
        _chk_tensor("z", z, torch.float32, (R, cfg.latent_dim), dev)
        if pi_eps is not None:
            _chk_tensor("pi_eps", pi_eps, torch.float32, (R, cfg.action_dim), dev)
        if qidx is not None:
            _chk_tensor("qidx", qidx, torch.int32, (2,), dev)
        tt, keep = self._task_tables(R, task_ids, task_emb_table, act_mask_table)
        action = torch.empty(R, cfg.action_dim, device=dev) if return_action else None
        q = torch.empty(R, device=dev)
        with torch.cuda.device(dev):
            self._check(self.lib.tdmpc2_plan_policy_value_mt(self._h, R, _ptr(z), tt, int(bool(use_target)), int(reduce == "min"),
                                                             _ptr(pi_eps), _ptr(qidx), C.c_uint64(int(seed) & (2**64 - 1)),
                                                             _ptr(action), _ptr(q), self._stream()))
        return action, q

    def td_target(self, next_z, reward, terminated, discount, pi_eps=None, qidx=None, seed: int = 0,
                  task_ids=None, task_emb_table=None, act_mask_table=None):
        """TDMPC2._td_target (tdmpc2.py:239-254) on flattened rows: next_z [R, L], reward / terminated [R] -> td [R].
        `discount`: python float (single task) or the per-task fp32 tensor TDMPC2.discount (multitask, tdmpc2.py:35-37)."""
        cfg, dev = self.cfg, self.device
        R = int(next_z.shape[0])
        _chk_tensor("next_z", next_z, torch.float32, (R, cfg.latent_dim), dev)
        _chk_tensor("reward", reward, torch.float32, (R,), dev)
        _chk_tensor("terminated", terminated, torch.float32, (R,), dev)
        if pi_eps is not None:
            _chk_tensor("pi_eps", pi_eps, torch.float32, (R, cfg.action_dim), dev)
        if qidx is not None:
            _chk_tensor("qidx", qidx, torch.int32, (2,), dev)
        disc_tab = discount if cfg.multitask else None
        tt, keep = self._task_tables(R, task_ids, task_emb_table, act_mask_table, disc_tab)
        td = torch.empty(R, device=dev)
        with torch.cuda.device(dev):
            self._check(self.lib.tdmpc2_plan_td_target_mt(self._h, R, _ptr(next_z), _ptr(reward), _ptr(terminated),
                                                          C.c_float(0.0 if cfg.multitask else float(discount)), tt,
                                                          _ptr(pi_eps), _ptr(qidx), C.c_uint64(int(seed) & (2**64 - 1)),
                                                          _ptr(td), self._stream()))
        return td

    # ------------------------------------------------------------------ packed weight file
    def export_packed(self) -> bytes:
        """Everything the binds produced (fragment-ordered weights, scales, LayerNorm parameters, encoder, target ensemble when
        bound) as one blob; specific to this handle's kernel family and arithmetic (tdmpc2_plan_export_packed)."""
        n = C.c_uint64()
        with torch.cuda.device(self.device):
            self._check(self.lib.tdmpc2_plan_packed_size(self._h, C.byref(n)))
            buf = (C.c_char * n.value)()
            self._check(self.lib.tdmpc2_plan_export_packed(self._h, C.cast(buf, C.c_void_p), n, self._stream()))
        return bytes(buf)

    def import_packed(self, blob: bytes, obs_dim: Optional[int] = None):
        """Restore the weights from `export_packed` output: host-to-device copies only (no packing kernels)."""
        buf = (C.c_char * len(blob)).from_buffer_copy(blob)
        with torch.cuda.device(self.device):
            self._check(self.lib.tdmpc2_plan_import_packed(self._h, C.cast(buf, C.c_void_p), C.c_uint64(len(blob)), self._stream()))
        if obs_dim is not None:
            self.obs_dim = int(obs_dim)

    def save_packed(self, path: str):
        with open(path, "wb") as f:
            f.write(self.export_packed())

    def load_packed(self, path: str, obs_dim: Optional[int] = None):
        with open(path, "rb") as f:
            self.import_packed(f.read(), obs_dim)

    # ------------------------------------------------------------------ planning
    def _common_inputs(self, E, z0, task_emb, act_mask, disc_pow):
        cfg, dev = self.cfg, self.device
        _chk_tensor("z0", z0, torch.float32, (E, cfg.latent_dim), dev)
        _chk_tensor("disc_pow", disc_pow, torch.float32, (E, cfg.horizon + 1), dev)
        if cfg.multitask:
            if task_emb is None or act_mask is None:
                raise ValueError("multitask planning needs task_emb and act_mask")
            _chk_tensor("task_emb", task_emb, torch.float32, (E, cfg.task_dim), dev)
            _chk_tensor("act_mask", act_mask, torch.float32, (E, cfg.action_dim), dev)

    def plan(self, z0, disc_pow, prev_mean, t0, eval_mode=False, task_emb=None, act_mask=None,
             tape: Optional[Dict[str, torch.Tensor]] = None, seed: int = 0, debug: bool = False,
             out: Optional[torch.Tensor] = None):
        """E plans in one call.  `prev_mean` [E,H,A] is updated in place.
        Returns action [E,A] (device), or (action, stages) when `debug`."""
        cfg, dev = self.cfg, self.device
        E = int(z0.shape[0])
        H, N, K, P, A, I = cfg.horizon, cfg.num_samples, cfg.num_elites, cfg.num_pi_trajs, cfg.action_dim, self.iterations
        self._common_inputs(E, z0, task_emb, act_mask, disc_pow)
        _chk_tensor("prev_mean", prev_mean, torch.float32, (E, H, A), dev)
        _chk_tensor("t0", t0, torch.uint8, (E,), dev)
        action = out if out is not None else torch.empty(E, A, device=dev, dtype=torch.float32)
        _chk_tensor("action", action, torch.float32, (E, A), dev)
        noise_p = None
        if tape is not None:
            noise = self._noise(tape, E)
            noise_p = C.byref(noise)
        dbg_p, stages = None, None
        if debug:
            NP = self._npad  # (stages come back sliced to the caller's num_samples)
            stages = {"value": torch.empty(E, I, NP, device=dev), "elite_idx": torch.empty(E, I, K, device=dev, dtype=torch.int32),
                      "score": torch.empty(E, I, K, device=dev), "mean": torch.empty(E, I, H, A, device=dev),
                      "std": torch.empty(E, I, H, A, device=dev), "actions": torch.empty(E, I, H, NP, A, device=dev)}
            dbg = Debug(**{k: v.data_ptr() for k, v in stages.items()})
            dbg_p = C.byref(dbg)
        with torch.cuda.device(dev):
            self._check(self.lib.tdmpc2_plan_run(self._h, E, _ptr(z0), _ptr(task_emb), _ptr(act_mask), _ptr(disc_pow),
                                                 _ptr(prev_mean), _ptr(t0), int(bool(eval_mode)), noise_p,
                                                 C.c_uint64(int(seed) & (2**64 - 1)), _ptr(action), dbg_p, self._stream()))
        if debug and self._npad != N:
            stages["value"] = stages["value"][:, :, :N].contiguous()
            stages["actions"] = stages["actions"][:, :, :, :N].contiguous()
        return (action, stages) if debug else action

    def estimate_value(self, z0, disc_pow, actions, pi_eps, qidx, task_emb=None, act_mask=None, trace=False):
        """TDMPC2._estimate_value (tdmpc2/tdmpc2.py:122-136) on given action sequences -> value [E,N].
        With `trace`, also returns (tiles [E*N/64, 5H+7, 64, L], scalars [E, N, H+2+A])."""
        cfg, dev = self.cfg, self.device
        self._whole_tiles_only("estimate_value")
        E = int(z0.shape[0])
        self._common_inputs(E, z0, task_emb, act_mask, disc_pow)
        _chk_tensor("actions", actions, torch.float32, (E, cfg.horizon, cfg.num_samples, cfg.action_dim), dev)
        _chk_tensor("pi_eps", pi_eps, torch.float32, (E, cfg.num_samples, cfg.action_dim), dev)
        _chk_tensor("qidx", qidx, torch.int32, (E, 2), dev)
        value = torch.empty(E, cfg.num_samples, device=dev, dtype=torch.float32)
        tiles = scalars = None
        if trace:
            if self.path == PATH_FUSED:  # the layered path dumps the per-row scalars only
                tiles = torch.zeros(E * cfg.num_samples // 64, 5 * cfg.horizon + 7, 64, cfg.latent_dim, device=dev)
            scalars = torch.zeros(E, cfg.num_samples, cfg.horizon + 2 + cfg.action_dim, device=dev)
        with torch.cuda.device(dev):
            self._check(self.lib.tdmpc2_plan_estimate_value_trace(
                self._h, E, _ptr(z0), _ptr(task_emb), _ptr(act_mask), _ptr(disc_pow), _ptr(actions), _ptr(pi_eps),
                _ptr(qidx), _ptr(value), _ptr(tiles), _ptr(scalars), self._stream()))
        return (value, tiles, scalars) if trace else value

    def refit(self, value, actions, act_mask=None):
        """Elite select + refit (tdmpc2/tdmpc2.py:184-197).  `value` [E,N] gets nan_to_num in place.
        Returns (mean, std, score, elite_idx)."""
        self._whole_tiles_only("refit")
        cfg, dev = self.cfg, self.device
        E = int(value.shape[0])
        H, N, K, A = cfg.horizon, cfg.num_samples, cfg.num_elites, cfg.action_dim
        _chk_tensor("value", value, torch.float32, (E, N), dev)
        _chk_tensor("actions", actions, torch.float32, (E, H, N, A), dev)
        mean = torch.empty(E, H, A, device=dev)
        std = torch.empty(E, H, A, device=dev)
        score = torch.empty(E, K, device=dev)
        idx = torch.empty(E, K, device=dev, dtype=torch.int32)
        with torch.cuda.device(dev):
            self._check(self.lib.tdmpc2_plan_refit(self._h, E, _ptr(value), _ptr(actions), _ptr(act_mask), _ptr(mean),
                                                   _ptr(std), _ptr(score), _ptr(idx), self._stream()))
        return mean, std, score, idx

    # ------------------------------------------------------------------ one plan sharded over ranks (tdmpc2_amd/dist.py)
    @property
    def shard_granularity(self) -> int:
        """Row ranges of shard_values are multiples of this many sample rows."""
        return 64 if self.path == PATH_FUSED else 128

    def shard_begin(self, z0, prev_mean, t0, task_emb=None, act_mask=None, tape=None, seed: int = 0):
        """Prologue of a plan whose sample rows are split over ranks: warm start + policy-prior trajectories
        (tdmpc2.py:154-170), replicated on every rank."""
        self._whole_tiles_only("shard_begin")
        cfg, dev = self.cfg, self.device
        E = int(z0.shape[0])
        _chk_tensor("z0", z0, torch.float32, (E, cfg.latent_dim), dev)
        _chk_tensor("prev_mean", prev_mean, torch.float32, (E, cfg.horizon, cfg.action_dim), dev)
        _chk_tensor("t0", t0, torch.uint8, (E,), dev)
        self._shard_noise = self._noise(tape, E) if tape is not None else None
        noise_p = C.byref(self._shard_noise) if self._shard_noise is not None else None
        with torch.cuda.device(dev):
            self._check(self.lib.tdmpc2_plan_shard_begin(self._h, E, _ptr(z0), _ptr(task_emb), _ptr(act_mask), _ptr(prev_mean),
                                                         _ptr(t0), noise_p, C.c_uint64(int(seed) & (2**64 - 1)), self._stream()))

    def shard_values(self, it: int, row_begin: int, row_end: int, z0, disc_pow, value, act_mask=None, seed: int = 0):
        """Sample the iteration's actions (all rows, replicated) and evaluate rows [row_begin, row_end) of every plan into
        value[E, N] (other columns untouched)."""
        cfg, dev = self.cfg, self.device
        E = int(z0.shape[0])
        _chk_tensor("value", value, torch.float32, (E, cfg.num_samples), dev)
        _chk_tensor("disc_pow", disc_pow, torch.float32, (E, cfg.horizon + 1), dev)
        noise_p = C.byref(self._shard_noise) if self._shard_noise is not None else None
        with torch.cuda.device(dev):
            self._check(self.lib.tdmpc2_plan_shard_values(self._h, E, int(it), int(row_begin), int(row_end), _ptr(z0), _ptr(act_mask),
                                                          _ptr(disc_pow), noise_p, C.c_uint64(int(seed) & (2**64 - 1)), _ptr(value),
                                                          self._stream()))

    def shard_refit(self, it: int, value, prev_mean, action, act_mask=None, eval_mode=False, seed: int = 0, stages=None):
        """Elite selection + refit on the complete value[E, N] (identical on every rank); the last iteration also picks the
        action and writes the new prev_mean."""
        cfg, dev = self.cfg, self.device
        E = int(value.shape[0])
        _chk_tensor("value", value, torch.float32, (E, cfg.num_samples), dev)
        _chk_tensor("action", action, torch.float32, (E, cfg.action_dim), dev)
        noise_p = C.byref(self._shard_noise) if self._shard_noise is not None else None
        dbg_p = None
        if stages is not None:
            dbg = Debug(**{k: v.data_ptr() for k, v in stages.items()})
            dbg_p = C.byref(dbg)
        with torch.cuda.device(dev):
            self._check(self.lib.tdmpc2_plan_shard_refit(self._h, E, int(it), _ptr(value), _ptr(act_mask), _ptr(prev_mean),
                                                         int(bool(eval_mode)), noise_p, C.c_uint64(int(seed) & (2**64 - 1)),
                                                         _ptr(action), dbg_p, self._stream()))

    def debug_buffers(self, E: int):
        """Stage buffers for the stage-wise entry points (shard_refit): sized to the handle's sample count, so whole tiles only."""
        self._whole_tiles_only("debug_buffers")
        cfg, dev, I = self.cfg, self.device, self.iterations
        H, N, K, A = cfg.horizon, cfg.num_samples, cfg.num_elites, cfg.action_dim
        return {"value": torch.empty(E, I, N, device=dev), "elite_idx": torch.empty(E, I, K, device=dev, dtype=torch.int32),
                "score": torch.empty(E, I, K, device=dev), "mean": torch.empty(E, I, H, A, device=dev),
                "std": torch.empty(E, I, H, A, device=dev), "actions": torch.empty(E, I, H, N, A, device=dev)}

    # ------------------------------------------------------------------ tuning / profiling
    def set_rows_per_workgroup(self, rows: int):
        """0 = automatic (32-row workgroups for calls with few plans: latency), or force 32 / 64 sample rows."""
        self._check(self.lib.tdmpc2_plan_set_tuning(self._h, 0, int(rows)))

    def set_fold_refit(self, mode):
        """Fused family: elite selection + refit inside the rollout launch (True / 1), as a launch of its own (False / 0),
        or chosen per call (2, the default: inside when the call fits the chip in one round of workgroups)."""
        self._check(self.lib.tdmpc2_plan_set_tuning(self._h, 1, int(mode)))

    def set_cluster(self, mode):
        """Fused family, split arithmetic: the single-plan latency path (8 workgroups per 32-row tile, cluster_kernels.cuh):
        0 never, 1 whenever all of a call's clusters fit the chip at once, 2 (default) = 1 plus, for a single non-episodic plan,
        a second cluster per tile that runs the reward chain beside the dynamics chain (cluster2_kernels.cuh)."""
        self._check(self.lib.tdmpc2_plan_set_tuning(self._h, 2, int(mode)))

    def set_fuse_ln(self, on):
        """Layered family, split arithmetic: LayerNorm + Mish / SimNorm + operand split inside the GEMM epilogue (1, default)
        or as a row kernel over fp32 pre-activations (0)."""
        self._check(self.lib.tdmpc2_plan_set_tuning(self._h, 3, int(bool(on))))

    def set_profiling(self, max_launches: int):
        """Bracket up to `max_launches` rollout-kernel launches with HIP events (0 = off)."""
        self._check(self.lib.tdmpc2_plan_set_profiling(self._h, int(max_launches)))

    def profile_read(self):
        ms, n = C.c_float(), C.c_int()
        self._check(self.lib.tdmpc2_plan_profile_read(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)
