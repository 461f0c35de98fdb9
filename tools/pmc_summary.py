"""Summarise rocprofv3 --pmc passes (one directory per pass, *_counter_collection.csv) per kernel and grid.

    python tools/pmc_summary.py gpurun_out/pmc_r01a > profiles/r01a_pmc.txt

Values are MEANS PER LAUNCH over the launches of (kernel, workgroup count).  FETCH_SIZE / WRITE_SIZE are in KiB
(rocprofv3 units); on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(/opt/skills/guides/MI355X_MICROARCH.md, HBM section) — the corrected figure is printed beside the raw one.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    only = sys.argv[2] if len(sys.argv) > 2 else "k_"
    agg = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for path in sorted(glob.glob(os.path.join(root, "*", "*_counter_collection.csv"))):
        with open(path) as f:
            seen = set()
            for r in csv.DictReader(f):
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
                if only not in name:
                    continue
                key = (name, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
                agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
                d = (r["Dispatch_Id"], path)
                if d not in seen:
                    seen.add(d)
                    dur[key].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    print(f"# source: {root}   (means per launch; durations are under counter collection, i.e. perturbed)")
    for key in sorted(agg, key=lambda k: (k[0], -k[1])):
        c = {n: sum(v) / len(v) for n, v in agg[key].items()}
        d = sum(dur[key]) / len(dur[key])
        print(f"\n{key[0]}  workgroups={key[1]}  launches={len(dur[key])}  mean_duration_us={d / 1e3:.1f}")
        for n in sorted(c):
            print(f"    {n:28s} {c[n]:18.1f}")
        g = c.get("GRBM_GUI_ACTIVE")
        if g:
            g = g / 8.0  # the counter is summed over the 8 XCDs
            print(f"    -> effective clock              {g / d:18.3f} GHz (GRBM_GUI_ACTIVE / 8 XCDs / duration)")
        if g and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            # MFMA-pipe busy cycles summed over the chip's 1024 SIMDs vs the cycles they had
            print(f"    -> MFMA pipe utilisation        {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (g * 1024):18.4f} "
                  f"(SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs))")
        if "SQ_WAVE_CYCLES" in c:
            w = c["SQ_WAVE_CYCLES"]
            for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if n in c:
                    print(f"    -> {n} / SQ_WAVE_CYCLES      {c[n] / w:10.4f}")
        if "FETCH_SIZE" in c:
            print(f"    -> HBM-side read bytes          {c['FETCH_SIZE'] * 1024 / 1e6:14.2f} MB raw, "
                  f"{2 * c['FETCH_SIZE'] * 1024 / 1e6:.2f} MB with the gfx950 x2 correction")
        if "WRITE_SIZE" in c:
            print(f"    -> HBM-side write bytes         {c['WRITE_SIZE'] * 1024 / 1e6:14.2f} MB (uncalibrated)")
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            print(f"    -> L2 hit rate                  {c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):14.4f}")


def emit_json(root, out_path, kernel="k_rollout"):
    """Per-launch HBM-side traffic of `kernel` by workgroup count -> small JSON that bench.py reads for
    `roofline.traffic` (bench.py cannot collect PMC counters from inside its own process)."""
    import json
    res = {}
    for path in sorted(glob.glob(os.path.join(root, "*", "*_counter_collection.csv"))):
        with open(path) as f:
            for r in csv.DictReader(f):
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()
                base = name.split("<")[0]
                # `kernel`: a base name ("ks_rollout": every instantiation) or a regular expression for the whole
                # instantiation name ("ks_rollout<\\d+, \\d, 8, 0, 0>": the split-arithmetic ones only -- the exact-fp32
                # instantiation runs on the same grid and would otherwise be averaged in)
                if "<" in kernel or "\\" in kernel:
                    import re
                    if not re.fullmatch(kernel, name):
                        continue
                elif base != kernel:
                    continue
                if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE"):
                    continue
                wgs = str(int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1))
                res.setdefault(wgs, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    out = {"kernel": kernel.split("<")[0], "match": kernel, "source": root, "note": "bytes per launch; fetch = FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read "
           "correction, MI355X_MICROARCH.md HBM section), write = WRITE_SIZE KiB x 1024 (uncalibrated)", "by_workgroups": {}}
    for wgs, c in res.items():
        f_ = sum(c.get("FETCH_SIZE", [0])) / max(len(c.get("FETCH_SIZE", [])), 1)
        w_ = sum(c.get("WRITE_SIZE", [0])) / max(len(c.get("WRITE_SIZE", [])), 1)
        out["by_workgroups"][wgs] = {"fetch_bytes": 2 * f_ * 1024, "write_bytes": w_ * 1024,
                                     "traffic_bytes": 2 * f_ * 1024 + w_ * 1024}
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    if "--json" in sys.argv:  # pmc_summary.py <dir> --json <out> [kernel]  |  pmc_summary.py --json <out> [kernel] <dir>
        i = sys.argv.index("--json")
        rest = sys.argv[1:i] + sys.argv[i + 2:]
        dirs = [a for a in rest if os.path.isdir(a)]
        names = [a for a in rest if not os.path.isdir(a)]
        if not dirs or i + 1 >= len(sys.argv):
            sys.exit("usage: pmc_summary.py <dir> --json <out.json> [kernel]")
        emit_json(dirs[0], sys.argv[i + 1], names[0] if names else "k_rollout")
    else:
        main()
