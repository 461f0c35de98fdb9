"""Summarise a rocprofv3 kernel trace (rocpd sqlite .db or *_kernel_trace.csv) per kernel AND launch grid.

    python tools/rocprof_summary.py gpurun_out/prof_r1/r1_results.db > profiles/r01_kernel_stats.txt

rocprofv3's own --stats table averages a kernel over all of its launches; the planner launches the
same kernel at very different grids (E=256 throughput steps, E=1 latency probe), so the per-grid
split is the number to compare with bench.py's live HIP-event timing.
"""
import csv
import sqlite3
import sys
from collections import defaultdict


def rows_from_db(path):
    cur = sqlite3.connect(path).cursor()
    q = ("select name, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, end - start "
         "from kernels")
    for name, gx, wx, lds, vg, ag, sg, dur in cur.execute(q):
        yield name, int(gx), int(wx), int(lds), int(vg), int(ag), int(sg), float(dur)


def rows_from_csv(path):
    with open(path) as f:
        for r in csv.DictReader(f):
            yield (r["Kernel_Name"], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]), int(r.get("LDS_Block_Size", 0)),
                   int(r.get("VGPR_Count", 0)), int(r.get("Accum_VGPR_Count", 0)), int(r.get("SGPR_Count", 0)),
                   float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))


def main():
    path = sys.argv[1]
    rows = rows_from_db(path) if path.endswith(".db") else rows_from_csv(path)
    agg = defaultdict(list)
    meta = {}
    for name, gx, wx, lds, vg, ag, sg, dur in rows:
        short = name.replace("(anonymous namespace)::", "")
        if short.startswith("void "):
            short = short[5:]
        short = short.split("(")[0]
        key = (short[:48], gx // max(wx, 1))
        agg[key].append(dur)
        meta[key] = (wx, lds, vg, ag, sg)
    total = sum(sum(v) for v in agg.values())
    print(f"# source: {path}")
    print(f"# {'kernel':48s} {'workgroups':>10s} {'wg_size':>7s} {'lds_B':>7s} {'vgpr':>5s} {'agpr':>5s} {'sgpr':>5s} "
          f"{'calls':>6s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for key, durs in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        wx, lds, vg, ag, sg = meta[key]
        s = sum(durs)
        print(f"  {key[0]:48s} {key[1]:10d} {wx:7d} {lds:7d} {vg:5d} {ag:5d} {sg:5d} {len(durs):6d} {s / 1e6:10.3f} "
              f"{s / len(durs) / 1e3:10.2f} {min(durs) / 1e3:10.2f} {max(durs) / 1e3:10.2f} {100 * s / total:6.2f}")


if __name__ == "__main__":
    main()
