"""CPU: the arithmetic of the f16x2-split MFMA mode (fused_kernels.cuh), emulated on the oracle network with the
kernels' operand scaling, is in the fp32 round-off class; the cheaper bf16x2 split is not (why it was rejected)."""
import pytest

from oracle import split_probe


@pytest.mark.parametrize("name", ["small", "c1"])
def test_f16x2_split_is_fp32_class(name):
    errs = split_probe.value_errors(name, modes=["fp32", "f16x2_3", "bf16x2_3"])
    print(name, errs)
    assert errs["f16x2_3"] < 3 * errs["fp32"] + 1e-6
    assert errs["f16x2_3"] < 2e-5
    assert errs["bf16x2_3"] > 3 * errs["f16x2_3"]
