"""Register / spill / LDS metadata of the kernels inside a built library (the gfx950 code object of the fat binary).

    python tools/kmeta.py [lib.so] [name filter ...]
"""
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(so):
    """The library holds one gfx950 code object per translation unit (tdmpc2_amd/csrc/build.sh): every ELF of the fat binary."""
    out = ""
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={d}/fatbin", so], check=True)
        data = open(f"{d}/fatbin", "rb").read()
        pos, n = data.find(b"\x7fELF"), 0
        while pos >= 0:
            nxt = data.find(b"\x7fELF", pos + 4)
            open(f"{d}/dev{n}.co", "wb").write(data[pos:nxt if nxt >= 0 else len(data)])
            r = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f"{d}/dev{n}.co"], capture_output=True, text=True)
            out += r.stdout
            pos, n = nxt, n + 1
    cur, res = {}, []
    keys = (".vgpr_count", ".agpr_count", ".sgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size",
            ".group_segment_fixed_size")
    for line in out.splitlines():
        line = line.strip()
        m = re.match(r"\.name:\s+(\S+)", line)
        if m:
            cur["name"] = m.group(1)
        for k in keys:
            m = re.match(re.escape(k) + r":\s+(\d+)", line)
            if m:
                cur[k[1:]] = int(m.group(1))
        if line.startswith(".wavefront_size"):
            res.append(cur)
            cur = {}
    return res


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines()


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else "tdmpc2_amd/libtdmpc2_plan.so"
    filt = [a for a in sys.argv[1:] if not a.endswith(".so")] or ["ks_rollout", "g_gemm"]
    ks = kernels(so)
    dn = demangle([k.get("name", "?") for k in ks])
    for k, n in zip(ks, dn):
        n = n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
        if any(f in n for f in filt):
            print(f"{n:44s} vgpr {k.get('vgpr_count', 0):4d} agpr {k.get('agpr_count', 0):4d} sgpr {k.get('sgpr_count', 0):4d} "
                  f"vspill {k.get('vgpr_spill_count', 0):4d} sspill {k.get('sgpr_spill_count', 0):4d} scratch {k.get('private_segment_fixed_size', 0):5d} "
                  f"lds {k.get('group_segment_fixed_size', 0):6d}")
