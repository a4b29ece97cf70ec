#!/usr/bin/env python3
"""Per-step busy time and idle gaps from a rocprofv3 `--kernel-trace --output-format csv` run: the kernels of the LAST
`steps` steps of a bench.py run (eager, or `--graph-child` replays), in start order -- sum of durations, sum of the gaps
between one kernel's end and the next one's start, and the kernels whose average duration differs most between two
traces.   python3 profiles/trace_gaps.py <trace dir A> [<trace dir B>] [--steps 5] [--marker bin_count_kernel]
A step is delimited by the first kernel of the path (`--marker`, launched once per step)."""
import csv
import glob
import sys
from collections import defaultdict


def load(d):
    files = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {d}")
    rows = []
    for r in csv.DictReader(open(files[0])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    return rows


def steps_of(rows, marker, steps):
    starts = [i for i, r in enumerate(rows) if marker in r[2]]
    if len(starts) < steps + 1:
        raise SystemExit(f"only {len(starts)} launches of {marker}")
    # whole steps only: from marker k to marker k+1
    segs = [rows[starts[k]:starts[k + 1]] for k in range(len(starts) - steps - 1, len(starts) - 1)]
    return segs


def summarize(d, marker, steps):
    segs = steps_of(load(d), marker, steps)
    busy = [sum(e - s for s, e, _ in seg) for seg in segs]
    span = [seg[-1][1] - seg[0][0] for seg in segs]
    gaps = [sum(max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)) for seg in segs]
    per = defaultdict(list)
    for seg in segs:
        for s, e, n in seg:
            per[n.split("(")[0][:90]].append(e - s)
    n = len(segs)
    print(f"{d}: {n} steps, {sum(len(s) for s in segs) / n:.1f} kernels per step (marker to marker, the next step's head excluded)")
    print(f"  busy {sum(busy) / n / 1e6:.3f} ms   gaps {sum(gaps) / n / 1e6:.3f} ms   span {sum(span) / n / 1e6:.3f} ms   mean gap {sum(gaps) / max(1, sum(len(s) - 1 for s in segs)) / 1e3:.2f} us")
    return {k: sum(v) / n for k, v in per.items()}


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    steps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 5
    marker = sys.argv[sys.argv.index("--marker") + 1] if "--marker" in sys.argv else "bin_count_kernel"
    a = summarize(args[0], marker, steps)
    if len(args) > 1:
        b = summarize(args[1], marker, steps)
        print(f"per-step kernel time, A vs B (us), largest differences first:")
        for k in sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, 0) - b.get(k, 0)))[:14]:
            print(f"  {a.get(k, 0) / 1e3:9.1f} {b.get(k, 0) / 1e3:9.1f} {(b.get(k, 0) - a.get(k, 0)) / 1e3:+8.1f}  {k}")
