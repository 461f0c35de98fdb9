"""MI355X-native TD-MPC2 planner: drop-in `TDMPC2.act()/plan()` over hand-written HIP kernels."""
from .config import Config, named_config, parse_cfg, planner_iterations  # noqa: F401

__all__ = ["Config", "named_config", "parse_cfg", "planner_iterations", "TDMPC2", "WorldModel", "NativePlanner"]


def __getattr__(name):  # lazy: importing the package must not require torch/HIP
    if name == "TDMPC2":
        from .tdmpc2 import TDMPC2
        return TDMPC2
    if name == "WorldModel":
        from .world_model import WorldModel
        return WorldModel
    if name == "NativePlanner":
        from .native import NativePlanner
        return NativePlanner
    raise AttributeError(name)
