"""Checkpoint key handling in the reference's layout.

A reference checkpoint is `{"model": state_dict}` (tdmpc2/tdmpc2.py:72-79).  Two
generations of Q-ensemble keys exist; `convert_state_dict` accepts both and
returns the current one, like the reference's loader does
(tdmpc2/common/layers.py:167-221), without needing tensordict.

  old:  _Qs.params.<n>, _target_Qs.params.<n>         n = 4*layer + {0 weight, 1 bias, 2 ln.weight, 3 ln.bias}
  new:  _Qs.params.<layer>.<name>, _detach_Qs_params.<layer>.<name>, _target_Qs_params.<layer>.<name>
        (+ tensordict meta entries  *.__batch_size / *.__device)
"""
from __future__ import annotations

from typing import Dict

import torch

_NAMES = ("weight", "bias", "ln.weight", "ln.bias")
META_SUFFIXES = ("__batch_size", "__device")


def is_meta_key(k: str) -> bool:
    return k.endswith(META_SUFFIXES)


def is_old_format(sd: Dict[str, torch.Tensor]) -> bool:
    """The reference's test (layers.py:172): a new-format dict carries `_detach_Qs_params.0.weight`; anything else that
    has flat-numbered Q keys is the old API."""
    if "_detach_Qs_params.0.weight" in sd:
        return False
    return any(k.startswith("_Qs.params.") and k[len("_Qs.params."):].isdigit() for k in sd)


def convert_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Return a new-format state dict without tensordict meta entries."""
    out = {}
    if "_detach_Qs_params.0.weight" in sd:  # already new format
        return {k: v for k, v in sd.items() if not is_meta_key(k)}
    for k, v in sd.items():
        if is_meta_key(k):
            continue
        if k.startswith("_Qs.params.") and k[len("_Qs.params."):].isdigit():
            n = int(k[len("_Qs.params."):])
            name = f"{n // 4}.{_NAMES[n % 4]}"
            out[f"_Qs.params.{name}"] = v
            out[f"_detach_Qs_params.{name}"] = v
        elif k.startswith("_target_Qs.params.") and k[len("_target_Qs.params."):].isdigit():
            n = int(k[len("_target_Qs.params."):])
            out[f"_target_Qs_params.{n // 4}.{_NAMES[n % 4]}"] = v
        else:
            out[k] = v
    # a checkpoint without the target / detach copies (e.g. synthetic weights): alias the online params
    for k in [k for k in out if k.startswith("_Qs.params.")]:
        tail = k[len("_Qs.params."):]
        out.setdefault(f"_detach_Qs_params.{tail}", out[k])
        out.setdefault(f"_target_Qs_params.{tail}", out[k])
    return out


def to_old_format(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Inverse mapping (used by tests to fabricate an old-style checkpoint)."""
    out = {}
    for k, v in sd.items():
        if is_meta_key(k) or k.startswith("_detach_Qs_params."):
            continue
        for new_p, old_p in (("_Qs.params.", "_Qs.params."), ("_target_Qs_params.", "_target_Qs.params.")):
            if k.startswith(new_p):
                layer, name = k[len(new_p):].split(".", 1)
                out[f"{old_p}{4 * int(layer) + _NAMES.index(name)}"] = v
                break
        else:
            out[k] = v
    return out
