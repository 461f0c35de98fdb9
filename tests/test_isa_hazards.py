"""tools/isa_hazards.py: the static scan for software-managed gfx940+ hazards around inline asm (round 5's K-split
failure was a 16-byte inline-asm store whose data registers the compiler overwrote one instruction later).  Synthetic
instruction streams pin every rule; the built library must scan clean."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))  # a real module: the library scan hands its work to child processes
import isa_hazards as hz  # noqa: E402


def _scan(lines):
    """lines of disassembly text -> rules reported"""
    dis = "0000000000001000 <kern>:\n" + "".join(
        f"\t{t:<60}// {0x1000 + 8 * i:012X}: 00000000\n" if not t.endswith(":") else f"{0x1000 + 8 * i:016x} <{t[:-1]}>:\n" for i, t in enumerate(lines))
    code, labels = hz.parse(dis)["kern"]
    return [r[1] for r in hz.scan("kern", code, labels)]


def test_store_data_hazard_is_the_round_5_bug():
    # what the compiler emitted behind the inline-asm store: one s_mov between the store and the write of its data registers
    bad = ["global_store_dwordx4 v[132:133], v[50:53], off sc1", "s_mov_b64 s[24:25], 0x2000", "v_lshl_add_u64 v[50:51], v[132:133], 0, s[24:25]"]
    assert _scan(bad) == ["store-data"]
    good = [bad[0], "s_nop 1", bad[1], bad[2]]
    assert _scan(good) == []
    assert _scan([bad[0], "s_mov_b64 s[24:25], 0x2000", "s_nop 0", bad[2]]) == []  # two wait states by any means
    # 64 bits of data or fewer: no hazard; a write of the ADDRESS registers: none either
    assert _scan(["global_store_dwordx2 v[132:133], v[50:51], off", "v_mov_b32 v50, 0"]) == []
    assert _scan(["global_store_dwordx4 v[132:133], v[50:53], off", "v_mov_b32 v132, 0"]) == []
    assert _scan(["buffer_store_dwordx4 v[4:7], v1, s[8:11], 0 offen", "v_mov_b32 v5, 0"]) == ["store-data"]
    assert _scan(["buffer_store_dwordx4 v[4:7], v1, s[8:11], 0 offen", "v_mov_b32 v1, 0"]) == []


def test_mfma_results_need_their_passes():
    mfma = "v_mfma_f32_32x32x16_f16 v[0:15], v[16:19], v[20:23], v[0:15]"
    assert _scan([mfma, "v_add_f32 v30, v0, v1"]) == ["mfma-read"]
    assert _scan([mfma, "s_nop 7", "v_add_f32 v30, v0, v1"]) == ["mfma-read"]            # 8 passes on gfx950: 12 wait states
    assert _scan([mfma, "s_nop 7", "s_nop 2", "v_add_f32 v30, v0, v1"]) == ["mfma-read"]  # 11
    assert _scan([mfma, "s_nop 7", "s_nop 3", "v_add_f32 v30, v0, v1"]) == []            # 12
    assert _scan([mfma, "global_store_dword v[40:41], v3, off"]) == ["mfma-read"]
    assert _scan([mfma, "ds_write_b32 v40, v3"]) == ["mfma-read"]
    # back-to-back accumulation into the same tuple is what the pipe is built for; A / B operands are not
    assert _scan([mfma, mfma]) == []
    assert _scan([mfma, "v_mfma_f32_32x32x16_f16 v[32:47], v[0:3], v[20:23], v[32:47]"]) == ["mfma-ab"]
    assert _scan([mfma, "v_mfma_f32_32x32x16_f16 v[8:23], v[40:43], v[44:47], v[8:23]"]) == ["mfma-ab"]  # overlapping, not the same C
    fp32 = "v_mfma_f32_32x32x2_f32 v[0:15], v16, v17, v[0:15]"  # 16 passes, fp32 inputs: 18
    assert _scan([fp32, "s_nop 7", "s_nop 7", "v_add_f32 v30, v0, v1"]) == ["mfma-read"]
    assert _scan([fp32, "s_nop 7", "s_nop 7", "s_nop 1", "v_add_f32 v30, v0, v1"]) == []


def test_valu_written_sgprs():
    assert _scan(["v_readfirstlane_b32 s4, v0", "global_load_dword v1, v2, s[4:5]"]) == ["sgpr-vmem"]
    assert _scan(["v_readfirstlane_b32 s4, v0", "s_nop 4", "global_load_dword v1, v2, s[4:5]"]) == []
    assert _scan(["s_mov_b32 s4, 0", "global_load_dword v1, v2, s[4:5]"]) == []  # SALU-written: interlocked
    assert _scan(["v_cmp_lt_f32 vcc, v0, v1", "v_div_fmas_f32 v2, v3, v4, v5"]) == ["vcc-divfmas"]
    assert _scan(["v_readfirstlane_b32 s7, v0", "v_readlane_b32 s8, v1, s7"]) == ["lane-select"]


def test_forwarding_hazards():
    assert _scan(["v_exp_f32 v1, v0", "v_add_f32 v2, v1, v1"]) == ["trans-fwd"]
    assert _scan(["v_exp_f32 v1, v0", "v_rcp_f32 v2, v1"]) == []
    assert _scan(["v_exp_f32 v1, v0", "s_nop 0", "v_add_f32 v2, v1, v1"]) == []
    assert _scan(["v_cvt_f32_f16_sdwa v1, v0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "v_add_f32 v2, v1, v1"]) == []
    assert _scan(["v_fma_mixhi_f16 v1, v0, v2, v3", "v_add_f32 v2, v1, v1"]) == ["dstsel-fwd"]


def test_branch_targets_are_followed():
    store = "global_store_dwordx4 v[132:133], v[50:53], off"
    assert _scan([store, "s_cbranch_scc1 L1", "s_endpgm", "L1:", "v_mov_b32 v50, 0"]) == ["store-data"]
    assert _scan([store, "s_nop 0", "s_cbranch_scc1 L1", "s_endpgm", "L1:", "v_mov_b32 v50, 0"]) == []


@pytest.fixture(scope="module")
def library_scan():
    if not os.path.exists(hz.OBJDUMP):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    lib = os.path.join(ROOT, "tdmpc2_amd", "libtdmpc2_plan.so")
    if not os.path.exists(lib):
        pytest.skip("library not built (python -c 'import __graft_entry__ as g; g.build()')")
    return hz.scan_library(lib, waits=True)


def test_the_built_library_scans_clean(library_scan):
    nk, ni, reports = library_scan
    assert nk >= 100 and ni > 1_000_000, (nk, ni)  # every family's code object was found and disassembled
    hazards = [r for r in reports if not r[1].endswith("-use")]
    assert not hazards, hazards[:5]


# kernels with a hand-written DMA / register ring whose counted waits depend on wave-uniform "is there a next slab" flags: the
# path-insensitive wait pass reports infeasible paths there (tools/isa_hazards.py); their schedules are covered by
# tests/test_ring_schedule.py (g_gemm_w, g_gemm_m) and by the parity suite
RING_KERNELS = ("ks_rollout", "ks_value", "ks_pitraj", "g_gemm_w", "g_gemm_m")


def test_no_load_result_is_used_before_its_wait_outside_the_hand_written_rings(library_scan):
    """Inline-asm loads are invisible to the compiler's wait insertion: the asm has to carry the wait (cl_ld16x2 / cl_ld16x8 of
    the cluster family, gw_ld4 / 8 / 12 of the K-split reduction -- whose first version did not, and summed registers that had
    not landed).  Every kernel without a flag-dependent ring must pass the counted-wait model outright; g_gemm_w's only
    register-destination loads are that reduction's, so it must be free of vmcnt reports too."""
    _, _, reports = library_scan
    waits = [r for r in reports if r[1].endswith("-use")]
    import re
    fam = lambda k: re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", k)  # noqa: E731
    outside = [r for r in waits if not fam(r[0]).startswith(RING_KERNELS)]
    assert not outside, outside[:5]
    assert all(fam(r[0]).startswith(RING_KERNELS) for r in waits)
    # (g_gemm_m has no register-destination global loads at all -- LDS-DMA only; with tools/variants/r6_reduce_scatter_epilogue.patch its
    # reduce-scatter loads, gm_ld_parts, carry their wait in the same statement)
    gw = [r for r in waits if fam(r[0]).startswith(("g_gemm_w", "g_gemm_m")) and r[1] == "vmcnt-use"]
    assert not gw, gw[:5]


def _waits(lines):
    dis = "0000000000001000 <kern>:\n" + "".join(
        f"\t{t:<60}// {0x1000 + 8 * i:012X}: 00000000\n" if not t.endswith(":") else f"{0x1000 + 8 * i:016x} <{t[:-1]}>:\n" for i, t in enumerate(lines))
    code, labels = hz.parse(dis)["kern"]
    return [r[1] for r in hz.scan_waits("kern", code, labels)]


def test_counted_waits_model():
    """the opt-in --waits pass: loads return in issue order per counter, scalar loads out of order"""
    ld = ["global_load_dwordx4 v[0:3], v20, s[4:5]", "global_load_dwordx4 v[4:7], v20, s[4:5] offset:1024"]
    assert _waits(ld + ["v_add_f32 v8, v0, v0"]) == ["vmcnt-use"]
    assert _waits(ld + ["s_waitcnt vmcnt(1)", "v_add_f32 v8, v0, v0"]) == []          # the older of two has landed
    assert _waits(ld + ["s_waitcnt vmcnt(1)", "v_add_f32 v8, v4, v4"]) == ["vmcnt-use"]
    assert _waits(ld + ["s_waitcnt vmcnt(0)", "v_add_f32 v8, v4, v4"]) == []
    assert _waits(ld + ["global_load_dwordx4 v[0:3], v20, s[4:5]"]) == []               # reload of the same registers: in order
    assert _waits(ld + ["global_load_dword v9, v0, s[4:5]"]) == ["vmcnt-use"]           # ... but not as an address
    assert _waits(["global_load_lds_dwordx4 v[2:3], off", "v_mov_b32 v2, 0"]) == []     # LDS-DMA: v[2:3] is the address
    assert _waits(["global_store_dwordx4 v[2:3], v[4:7], off", "global_load_dword v9, v20, s[4:5]", "s_waitcnt vmcnt(1)",
                   "v_mov_b32 v10, v9"]) == ["vmcnt-use"]                                # the store counts, the load is the newest
    assert _waits(["ds_read_b128 v[0:3], v20", "ds_read_b128 v[4:7], v20 offset:16", "s_waitcnt lgkmcnt(1)", "v_add_f32 v8, v0, v1"]) == []
    assert _waits(["s_load_dwordx2 s[4:5], s[0:1], 0x0", "ds_read_b32 v1, v20", "s_waitcnt lgkmcnt(1)", "s_add_u32 s6, s4, 1"]) == ["lgkmcnt-use"]
    assert _waits(["s_load_dwordx2 s[4:5], s[0:1], 0x0", "s_waitcnt lgkmcnt(0)", "s_add_u32 s6, s4, 1"]) == []
    # a loop: the second trip sees what the first left in flight
    assert _waits(["L0:", "v_add_f32 v8, v0, v0", "global_load_dwordx4 v[0:3], v20, s[4:5]", "s_cbranch_scc1 L0", "s_waitcnt vmcnt(0)"]) == ["vmcnt-use"]
