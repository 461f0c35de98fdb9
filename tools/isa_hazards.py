#!/usr/bin/env python3
"""Static scan of the built library's gfx950 machine code for the software-managed hazards of gfx940+ that the compiler
cannot see when ONE SIDE OF THE PAIR IS INLINE ASM (its hazard recognizer handles its own instructions, not the text of
an asm statement).  Written after round 5's K-split failure: a 16-byte inline-asm store whose data registers the compiler
overwrote one instruction later.  The scan works on the final instruction stream of every kernel, whoever
emitted the instructions.

    python tools/isa_hazards.py [--waits] [path/to/libtdmpc2_plan.so | x.hsaco]        exit code 1 if anything is reported

Rules (wait states: every instruction counts one, `s_nop N` counts N + 1; gfx940 / gfx950 figures as LLVM's
GCNHazardRecognizer applies them to compiler-emitted code):
  store-data   VMEM / FLAT store of more than 64 bits of data -> a VALU write of one of its data registers: 2
  mfma-read    MFMA (XDL) writes VGPRs -> non-MFMA VALU read or write, VMEM / DS / FLAT read of them: passes + 4 (gfx950, passes
               != 2; passes + 3 for 2 passes); fp32-input MFMA ("SMFMA"): passes + 2
  mfma-ab      MFMA writes VGPRs -> MFMA reads them as A or B, or as a C that overlaps without being the same tuple: same figures
  sgpr-vmem    VALU writes an SGPR / VCC -> VMEM reads it (resource, offset or saddr): 5
  lane-select  VALU writes an SGPR -> v_readlane / v_writelane uses it as the lane select: 4
  vcc-divfmas  VALU writes VCC -> v_div_fmas: 4
  trans-fwd    transcendental VALU result -> read by the next non-transcendental VALU instruction: 1
  dstsel-fwd   VALU write of half a register (op_sel high half, SDWA, *_mixhi, cvt_*_sr / pk hi forms) -> VALU read: 1
With --waits also (PATH-INSENSITIVE: a wait under a wave-uniform branch is taken as the branch falls, so the hand-written
rings, whose counted waits depend on "is there a next slab" flags, produce reports that a person has to read; every kernel
without such a ring -- 1.5 M compiler-scheduled instructions -- scans clean, which is what validates the counter model):
  vmcnt-use    a register that a VMEM load / returning atomic still in flight will write is read or written before an
               `s_waitcnt vmcnt(N)` small enough to cover it (returns are in issue order; stores and LDS-DMA loads count too)
  lgkmcnt-use  the same for DS reads (in order) and scalar memory loads (out of order: only lgkmcnt(0) covers them)
               -- the compiler inserts these waits for its own loads; an inline-asm load is invisible to it, so the asm (or the
               code around it) must carry the wait: a too-weak counted wait in a hand-written ring is exactly the kind of bug that
               passes most runs
Branches: the scan follows the fall-through path and, at every branch, also the first instructions of the target with the
state it has there.  It does not prove absence (indirect control flow, waits across calls); it finds the straight-line cases,
which is where inline asm sits.
"""
import os
import re
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
TRANS = ("v_exp_", "v_log_", "v_rcp_", "v_rsq_", "v_sqrt_", "v_sin_", "v_cos_")
REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\b(vcc|exec|m0)\b")


def regs(op):
    """register units named by one operand: {('v', n), ('s', n), ('vcc',), ...}"""
    out = set()
    for m in REG.finditer(op):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        elif m.group(3):
            for i in range(int(m.group(4)), int(m.group(5)) + 1):
                out.add((m.group(3), i))
        else:
            out.add((m.group(6),))
    return out


def mfma_passes(mn):
    m = re.match(r"v_mfma_(?:f32|i32|f64)_(\d+)x(\d+)x(\d+)", mn)
    if not m:
        return 4
    a, b, k = int(m.group(1)), int(m.group(2)), int(m.group(3))
    f32_in = mn.endswith("_f32") and "xf32" not in mn and not mn.endswith("f16_f32")
    if mn.endswith("x2_f32") or mn.endswith("x1_f32") or mn.endswith("x4_f32") and a == 16:
        f32_in = True
    if a == 32:
        return 16 if f32_in else 8
    if a == 16:
        return 8 if f32_in else 4
    return 2


class Ins:
    __slots__ = ("addr", "mn", "ops", "text", "dst", "src", "kind", "wait")

    def __init__(self, addr, text):
        self.addr, self.text = addr, text
        parts = text.split(None, 1)
        self.mn = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        self.ops = [o.strip() for o in rest.split(",")] if rest else []
        mn = self.mn
        self.wait = 1
        if mn == "s_nop":
            self.wait = int(self.ops[0], 0) + 1
        self.kind = ("mfma" if mn.startswith("v_mfma") or mn.startswith("v_smfmac") else "valu" if mn.startswith("v_") else
                     "vmem" if mn.split("_")[0] in ("global", "buffer", "flat", "scratch") else "ds" if mn.startswith("ds_") else
                     "salu" if mn.startswith("s_") else "other")
        dst, src = set(), set()
        ops = self.ops
        if self.kind in ("valu", "mfma"):
            if mn.startswith("v_cmpx"):
                dst.add(("exec",))
                for o in ops:
                    src |= regs(o)
            elif mn.startswith("v_swap") or mn.startswith("v_permlane") and "swap" in mn:
                for o in ops[:2]:
                    dst |= regs(o)
                    src |= regs(o)
            else:
                ndst = 1
                if re.match(r"v_(add|sub|subrev)_co_|v_(addc|subb|subbrev)_co_|v_div_scale|v_mad_(u64_u32|i64_i32)", mn):
                    ndst = 2
                for o in ops[:ndst]:
                    dst |= regs(o)
                for o in ops[ndst:]:
                    src |= regs(o)
                if mn.startswith("v_cmp") and not ops[0].startswith(("s", "vcc")):
                    dst.add(("vcc",))  # VOPC form: implicit VCC
                if re.match(r"v_(addc|subb|subbrev|cndmask)", mn) and len(ops) == 3:
                    src.add(("vcc",))
                if mn.startswith("v_div_fmas"):
                    src.add(("vcc",))
        elif self.kind == "vmem":
            if "_store_" in mn or "_atomic_" in mn and not any(" sc0" in " " + o or o.endswith("glc") for o in ops):
                for o in ops:
                    src |= regs(o)
            elif " lds" in " " + " ".join(ops) or "_lds_" in mn:  # LDS-DMA: no register destination
                for o in ops:
                    src |= regs(o)
            else:
                dst |= regs(ops[0]) if ops else set()
                for o in ops[1:]:
                    src |= regs(o)
        elif self.kind == "ds":
            if mn.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append")) or "_rtn" in mn:
                dst |= regs(ops[0]) if ops else set()
                for o in ops[1:]:
                    src |= regs(o)
            else:
                for o in ops:
                    src |= regs(o)
        elif self.kind == "salu":
            if mn.startswith(("s_cmp", "s_bitcmp", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_endpgm",
                              "s_setprio", "s_sendmsg", "s_setreg", "s_trap", "s_icache", "s_dcache")):
                for o in ops:
                    src |= regs(o)
            else:
                dst |= regs(ops[0]) if ops else set()
                for o in ops[1:]:
                    src |= regs(o)
        self.dst, self.src = dst, src

    def store_data(self):
        """data registers of a store of more than 64 bits"""
        if self.kind != "vmem" or "_store_" not in self.mn or not re.search(r"dwordx[34]$", self.mn):
            return set()
        # global / flat / scratch: vaddr, vdata, saddr; buffer: vdata, vaddr, srsrc, soffset
        return regs(self.ops[0] if self.mn.startswith("buffer") else self.ops[1])

    def half_write(self):
        t = self.text
        if self.kind != "valu":
            return False
        if "dst_sel:" in t:
            return "dst_sel:DWORD" not in t
        return "mixhi" in self.mn or re.search(r"op_sel:\[[01],[01],?[01]?,1\]", t) is not None


def parse(dis):
    """-> {kernel: ([Ins], {label: index})}"""
    kernels, cur, name = {}, None, None
    for line in dis.splitlines():
        m = re.match(r"^([0-9a-f]+) <(.+)>:$", line)
        if m:
            lab = m.group(2)
            if not re.fullmatch(r"L\d+", lab) or cur is None:
                name = lab
                cur = kernels.setdefault(name, ([], {}))
            cur[1][lab] = len(cur[0])
            continue
        m = re.match(r"^\s+(\S.*?)\s*//\s*([0-9A-Fa-f]+):", line)
        if m and cur is not None:
            cur[0].append(Ins(int(m.group(2), 16), m.group(1).strip()))
    return kernels


class Pending:
    """producers whose hazard window is still open: (rule, regs, wait states left, producer)"""

    def __init__(self, items=None):
        self.items = list(items or [])

    def copy(self):
        return Pending(self.items)

    def advance(self, n):
        self.items = [(r, g, w - n, p) for (r, g, w, p) in self.items if w - n > 0]

    def add(self, rule, regset, wait, prod):
        if regset:
            self.items.append((rule, frozenset(regset), wait, prod))


def check(ins, pend, prev, found, kernel):
    def hit(rule, prod, left, regset):
        found.append((kernel, rule, prod, ins, left, sorted(regset)[:4]))

    for rule, g, left, prod in pend.items:
        if rule == "store-data":
            if ins.kind in ("valu", "mfma") and ins.dst & g:
                hit(rule, prod, left, ins.dst & g)
        elif rule == "mfma":
            if ins.kind == "mfma":
                ab = set()
                for o in ins.ops[1:3]:
                    ab |= regs(o)
                c = regs(ins.ops[3]) if len(ins.ops) > 3 else set()
                if ab & g:
                    hit("mfma-ab", prod, left, ab & g)
                elif c & g and c != set(g):
                    hit("mfma-ab", prod, left, c & g)
            elif ins.kind == "valu" and (ins.src | ins.dst) & g:
                hit("mfma-read", prod, left, (ins.src | ins.dst) & g)
            elif ins.kind in ("vmem", "ds") and ins.src & g:
                hit("mfma-read", prod, left, ins.src & g)
        elif rule == "sgpr-vmem":
            if ins.kind == "vmem" and ins.src & g:
                hit(rule, prod, left, ins.src & g)
        elif rule == "lane-select":
            if ins.mn.startswith(("v_readlane", "v_writelane")) and regs(ins.ops[-1]) & g:
                hit(rule, prod, left, regs(ins.ops[-1]) & g)
        elif rule == "vcc-divfmas":
            if ins.mn.startswith("v_div_fmas"):
                hit(rule, prod, left, g)
        elif rule == "trans-fwd":
            if ins.kind in ("valu", "mfma") and not ins.mn.startswith(TRANS) and ins.src & g:
                hit(rule, prod, left, ins.src & g)
        elif rule == "dstsel-fwd":
            if ins.kind in ("valu", "mfma") and ins.src & g:
                hit(rule, prod, left, ins.src & g)


def produce(ins, pend):
    sd = ins.store_data()
    if sd:
        pend.add("store-data", sd, 2, ins)
    if ins.kind == "mfma":
        p = mfma_passes(ins.mn)
        f32_in = ins.mn.endswith("_f32") and p in (8, 16) and re.search(r"x(1|2|4)_f32$", ins.mn) is not None
        pend.add("mfma", {r for r in ins.dst if r[0] in "va"}, p + 2 if f32_in else (p + 3 if p == 2 else p + 4), ins)
    if ins.kind == "valu":
        sg = {r for r in ins.dst if r[0] in ("s", "vcc")}
        if sg:
            pend.add("sgpr-vmem", sg, 5, ins)
            pend.add("lane-select", {r for r in sg if r[0] == "s"}, 4, ins)
            if ("vcc",) in sg:
                pend.add("vcc-divfmas", {("vcc",)}, 4, ins)
        vd = {r for r in ins.dst if r[0] == "v"}
        if ins.mn.startswith(TRANS):
            pend.add("trans-fwd", vd, 1, ins)
        if ins.half_write():
            pend.add("dstsel-fwd", vd, 1, ins)


def scan(kernel, code, labels):
    found = []

    def run(i, pend, budget, follow):
        while i < len(code) and (budget is None or budget > 0):
            ins = code[i]
            check(ins, pend, None, found, kernel)
            pend.advance(ins.wait)
            produce(ins, pend)
            if budget is not None:
                budget -= ins.wait
            if follow and ins.mn.startswith(("s_cbranch", "s_branch")) and pend.items:
                tgt = ins.ops[0].split()[0] if ins.ops else ""
                m = re.search(r"<(.+?)(\+0x[0-9a-f]+)?>", ins.text)
                if m and m.group(1) in labels and not m.group(2):
                    run(labels[m.group(1)], pend.copy(), 24, False)
                elif tgt in labels:
                    run(labels[tgt], pend.copy(), 24, False)
            if ins.mn in ("s_endpgm", "s_branch", "s_setpc_b64"):
                pend = Pending()
            i += 1

    run(0, Pending(), None, True)
    # one report per (producer, consumer, rule)
    seen, out = set(), []
    for f in found:
        key = (f[1], f[2].addr, f[3].addr)
        if key not in seen:
            seen.add(key)
            out.append(f)
    return out


WAITCNT = re.compile(r"(vmcnt|lgkmcnt|expcnt)\((\d+)\)")


class Waits:
    """memory operations in flight: vm = [(regs the op will write, Ins)], lgkm = [(kind 'ds' | 'smem', regs, Ins)]"""

    def __init__(self, vm=None, lgkm=None):
        self.vm, self.lgkm = list(vm or []), list(lgkm or [])

    def copy(self):
        return Waits(self.vm, self.lgkm)


def wait_step(ins, w, found, kernel):
    touched = ins.src | ins.dst
    if touched:
        for q, rule in ((w.vm, "vmcnt-use"), (w.lgkm, "lgkmcnt-use")):
            # a second load into the same registers behind the first needs no wait: returns of one counter are in order
            same_q = (rule == "vmcnt-use" and ins.kind == "vmem") or (rule == "lgkmcnt-use" and ins.kind == "ds")
            look = ins.src if same_q else touched
            for k, ent in enumerate(q):
                g = ent[-2]
                if rule == "lgkmcnt-use" and same_q and ent[0] == "smem":
                    g = g and (touched & g)
                if g and look & g:
                    found.append((kernel, rule, ent[-1], ins, len(q) - k, sorted(touched & g)[:4]))
                    q[k] = ent[:-2] + (frozenset(), ent[-1])  # reported once
    mn = ins.mn
    if mn == "s_waitcnt":
        m = WAITCNT.findall(ins.text)
        if not m and ins.ops:  # raw immediate: treat as a wait for everything
            m = [("vmcnt", "0"), ("lgkmcnt", "0")]
        for name, n in m:
            n = int(n)
            if name == "vmcnt":
                w.vm = w.vm[len(w.vm) - n:] if n else []
            elif name == "lgkmcnt":
                if n == 0:
                    w.lgkm = []
                elif not any(e[0] == "smem" for e in w.lgkm):
                    w.lgkm = w.lgkm[len(w.lgkm) - n:]
        return
    if ins.kind == "vmem":
        w.vm.append((frozenset(r for r in ins.dst if r[0] in "va"), ins))
        if mn.startswith("flat_"):
            w.lgkm.append(("smem", frozenset(r for r in ins.dst if r[0] in "va"), ins))  # may return through either path, out of order
    elif ins.kind == "ds":
        w.lgkm.append(("ds", frozenset(r for r in ins.dst if r[0] in "va"), ins))
    elif mn.startswith(("s_load_", "s_buffer_load_", "s_memtime", "s_memrealtime", "s_scratch_load")):
        w.lgkm.append(("smem", frozenset(ins.dst), ins))
    elif mn.startswith(("s_sendmsg", "s_store_", "s_dcache", "s_atc_probe")):
        w.lgkm.append(("smem", frozenset(), ins))


def scan_waits(kernel, code, labels, follow_limit=256):
    found = []

    def target(ins):
        m = re.search(r"\b(L\d+)\b", ins.text)
        return labels.get(m.group(1)) if m else None

    def run(i, w, budget, follow):
        while i < len(code) and (budget is None or budget > 0):
            ins = code[i]
            wait_step(ins, w, found, kernel)
            if budget is not None:
                budget -= 1
            if ins.mn.startswith(("s_cbranch", "s_branch")):
                t = target(ins)
                if follow and t is not None and (w.vm or w.lgkm):
                    run(t, w.copy(), follow_limit, False)
                if ins.mn == "s_branch":
                    if follow:
                        w.vm, w.lgkm = [], []  # the fall-through below is reached from elsewhere
                    elif t is not None:
                        i = t  # a followed path takes the jump
                        continue
                    else:
                        return
            if ins.mn in ("s_endpgm", "s_setpc_b64", "s_swappc_b64"):
                if not follow:
                    return
                w.vm, w.lgkm = [], []
            i += 1

    run(0, Waits(), None, True)
    seen, out = set(), []
    for f in found:
        key = (f[1], f[2].addr, f[3].addr)
        if key not in seen:
            seen.add(key)
            out.append(f)
    return out


def code_objects(lib, tmp):
    """the gfx950 code objects bundled in a host library"""
    cp = os.path.join(tmp, "lib.so")
    with open(lib, "rb") as f, open(cp, "wb") as g:
        g.write(f.read())
    subprocess.run([OBJDUMP, "--offloading", cp], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
    return sorted(os.path.join(tmp, n) for n in os.listdir(tmp) if "amdgcn" in n)


def scan_object(co, waits=False):
    dis = subprocess.run([OBJDUMP, "-d", "--mcpu=gfx950", "--symbolize-operands", co], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                         text=True).stdout
    nk = ni = 0
    reports = []
    for name, (code, labels) in parse(dis).items():
        if code:
            nk += 1
            ni += len(code)
            reports += scan(name, code, labels)
            if waits:
                reports += scan_waits(name, code, labels)
    # plain tuples: the result crosses a process boundary
    return nk, ni, [(k, rule, prod.addr, prod.text, cons.addr, cons.text, left, rg) for (k, rule, prod, cons, left, rg) in reports]


def scan_library(lib, jobs=None, waits=False):
    """-> (kernels, instructions, [(kernel, rule, producer address, producer, consumer address, consumer, wait states short, registers)])"""
    from concurrent.futures import ProcessPoolExecutor

    with tempfile.TemporaryDirectory() as tmp:
        objs = [lib] if lib.endswith((".co", ".hsaco")) else code_objects(lib, tmp)
        if not objs:
            raise RuntimeError(f"no gfx950 code object found in {lib}")
        with ProcessPoolExecutor(max_workers=jobs or min(len(objs), os.cpu_count() or 1)) as ex:
            parts = list(ex.map(scan_object, objs, [waits] * len(objs)))
    return sum(p[0] for p in parts), sum(p[1] for p in parts), [r for p in parts for r in p[2]]


def main(argv):
    waits = "--waits" in argv
    argv = [a for a in argv if a != "--waits"]
    lib = argv[1] if len(argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tdmpc2_amd", "libtdmpc2_plan.so")
    nk, ni, reports = scan_library(lib, waits=waits)
    print(f"{nk} kernels, {ni} instructions scanned")
    by = {}
    for r in reports:
        by.setdefault((r[0], r[1]), []).append(r)
    for (kernel, rule), rs in sorted(by.items()):
        print(f"\n{rule}: {len(rs)} in {kernel}")
        for _, _, pa, pt, ca, ct, left, rg in rs[:6]:
            what = "still in flight, position from the newest" if rule.endswith("-use") else "wait states short"
            print(f"    {pa:08x}  {pt}\n    {ca:08x}  {ct}      <- {what}: {left}, registers {rg}")
    print(f"\n{len(reports)} potential hazard(s)")
    return 1 if reports else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
