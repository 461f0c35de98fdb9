"""Per-kernel digest of the device code inside a built library: a refactor that is meant to change nothing can be checked here,
without a GPU -- identical digests = identical machine code.

    python tools/isa_digest.py tdmpc2_amd/libtdmpc2_plan.so > /tmp/before.txt     # ... edit, rebuild ...
    python tools/isa_digest.py tdmpc2_amd/libtdmpc2_plan.so | diff /tmp/before.txt -
"""
import hashlib
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def digests(so):
    res = {}
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={d}/fatbin", so], check=True)
        data = open(f"{d}/fatbin", "rb").read()
        pos, n = data.find(b"\x7fELF"), 0
        while pos >= 0:
            nxt = data.find(b"\x7fELF", pos + 4)
            open(f"{d}/dev{n}.co", "wb").write(data[pos:nxt if nxt >= 0 else len(data)])
            out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-leading-addr", f"{d}/dev{n}.co"], capture_output=True, text=True).stdout
            cur, h = None, None
            for line in out.splitlines():
                m = re.match(r"^[0-9a-f]* ?<(.+)>:$", line.strip())
                if m:
                    if cur:
                        res[cur] = h.hexdigest()[:16]
                    cur, h = m.group(1), hashlib.sha1()
                elif cur:
                    # the encoding bytes (after "//") are position independent except for the address comment in front of them
                    enc = line.split("//")[-1].split(":")[-1].strip() if "//" in line else line.strip()
                    h.update(enc.encode())
            if cur:
                res[cur] = h.hexdigest()[:16]
            pos, n = nxt, n + 1
    return res


if __name__ == "__main__":
    so = sys.argv[1] if len(sys.argv) > 1 else "tdmpc2_amd/libtdmpc2_plan.so"
    r = digests(so)
    names = subprocess.run(["c++filt"], input="\n".join(r), capture_output=True, text=True).stdout.splitlines()
    for mangled, name in sorted(zip(r, names), key=lambda x: x[1]):
        print(r[mangled], name.replace("(anonymous namespace)::", "")[:150])
