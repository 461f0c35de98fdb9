"""Offline look at a kernel timeline (tools/gpu_r4v.sh: start_ns,end_ns,queue,grid,wg,kernel): how much of a planning stage the
GPU is idle, runs one kernel, or runs two; which kernels the time goes to.

    python tools/timeline.py gpurun_out/r4v_c3_timeline.csv.gz 30      # stages delimited by k_refit launches of grid <envs>
"""
import collections
import csv
import gzip
import sys


def load(path):
    f = gzip.open(path, "rt") if path.endswith(".gz") else open(path)
    rows = [dict(r, start=int(r["start_ns"]), end=int(r["end_ns"]), grid=int(r["grid"])) for r in csv.DictReader(f)]
    rows.sort(key=lambda r: r["start"])
    return rows


def depth_profile(rows, t0, t1):
    """ns spent with 0, 1, 2, >= 3 kernels running inside [t0, t1)."""
    ev = []
    for r in rows:
        s, e = max(r["start"], t0), min(r["end"], t1)
        if e > s:
            ev.append((s, 1))
            ev.append((e, -1))
    ev.sort()
    prof = collections.Counter()
    d, last = 0, t0
    for t, k in ev:
        prof[min(d, 3)] += t - last
        last = t
        d += k
    prof[min(d, 3)] += t1 - last
    return prof


def main():
    path, envs = sys.argv[1], int(sys.argv[2])
    rows = load(path)
    refits = [i for i, r in enumerate(rows) if r["kernel"].startswith("k_refit") and r["grid"] == envs]
    # runs of consecutive stages: refit-to-refit distances that look like a stage (no host gap between steps beyond 3x median)
    gaps = [rows[refits[i + 1]]["end"] - rows[refits[i]]["end"] for i in range(len(refits) - 1)]
    med = sorted(gaps)[len(gaps) // 2]
    print(f"{len(refits)} refits of grid {envs}; median refit-to-refit {med / 1e3:.1f} us")
    stages = [(refits[i], refits[i + 1]) for i in range(len(refits) - 1) if gaps[i] < 1.5 * med]
    tot = collections.Counter()
    per_kernel = collections.defaultdict(lambda: [0, 0.0])
    span = 0
    for a, b in stages:
        t0, t1 = rows[a]["end"], rows[b]["end"]
        sub = [r for r in rows[a + 1:b + 1]]
        tot.update(depth_profile(sub, t0, t1))
        span += t1 - t0
        for r in sub:
            k = (r["kernel"], r["grid"])
            per_kernel[k][0] += 1
            per_kernel[k][1] += r["end"] - r["start"]
    n = len(stages)
    print(f"{n} stages, mean {span / n / 1e3:.1f} us; time with 0 / 1 / 2 / >=3 kernels running: "
          + " / ".join(f"{tot[d] / n / 1e3:.1f} us ({100.0 * tot[d] / span:.1f} %)" for d in range(4)))
    print(f"{'kernel':44s} {'grid':>6s} {'per stage':>9s} {'mean us':>9s} {'sum us/stage':>12s}")
    for (k, g), (c, ns) in sorted(per_kernel.items(), key=lambda kv: -kv[1][1]):
        print(f"{k[:44]:44s} {g:6d} {c / n:9.2f} {ns / c / 1e3:9.1f} {ns / n / 1e3:12.1f}")
    print(f"sum of kernel durations per stage {sum(v[1] for v in per_kernel.values()) / n / 1e3:.1f} us")
    if len(sys.argv) > 3:  # dump one stage
        a, b = stages[len(stages) // 2]
        t0 = rows[a]["end"]
        for r in rows[a + 1:b + 1]:
            print(f"{(r['start'] - t0) / 1e3:9.1f} {(r['end'] - t0) / 1e3:9.1f} q{r['queue']} {r['grid']:5d} {r['kernel'][:50]}")


if __name__ == "__main__":
    main()
