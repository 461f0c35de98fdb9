"""CPU tests of the measurement helpers whose output ends up under profiles/ (tools/timeline.py, tools/isa_digest.py,
tools/pmc_summary.py's corrections are covered where they are used): a wrong tool is a wrong number in DESIGN.md."""
import gzip
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_timeline_depth_profile_and_stage_table(tmp_path):
    """Two queues: kernel A 0-100, kernel B 50-150, idle 150-160, refit 160-170 -> idle 10, one kernel 100 + 10, two 50."""
    import timeline

    rows = ["start_ns,end_ns,queue,grid,wg,kernel"]
    t = 0
    for _ in range(4):  # four identical stages, each closed by a k_refit launch of grid 30
        rows += [f'{t},{t + 100},1,448,512,"g_gemm_w<1>"', f'{t + 50},{t + 150},2,448,512,"g_gemm_w<1>"',
                 f'{t + 160},{t + 170},1,30,256,"k_refit"']
        t += 170
    p = tmp_path / "tl.csv.gz"
    with gzip.open(p, "wt") as f:
        f.write("\n".join(rows) + "\n")
    r = timeline.load(str(p))
    assert len(r) == 12 and r[0]["start"] == 0 and r[-1]["kernel"] == "k_refit"
    prof = timeline.depth_profile(r[3:6], 170, 340)  # second stage, from the first refit's end to the second's
    # window [170, 340): A 170-270, B 220-320, refit 330-340
    assert prof[0] == 10 and prof[2] == 50 and prof[1] == 50 + 50 + 10
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "timeline.py"), str(p), "30"], capture_output=True, text=True, check=True).stdout
    assert "3 stages" in out and "g_gemm_w<1>" in out and "2.00" in out  # two GEMMs per stage


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="needs the ROCm LLVM tools")
def test_isa_digest_sees_every_kernel_family():
    import isa_digest

    so = os.path.join(ROOT, "tdmpc2_amd", "libtdmpc2_plan.so")
    if not os.path.exists(so):
        pytest.skip("library not built")
    a = isa_digest.digests(so)
    assert len(a) > 100
    names = " ".join(a)
    for frag in ("ks_rollout", "ks_rollout_cl2", "g_gemm_w", "g_gemm_s", "refit", "k_encode"):
        assert frag in names, frag
    assert len(set(a.values())) > 0.9 * len(a)  # distinct kernels, distinct code
