#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats --output-format csv` run into a small text table
(per-kernel calls / total / average / share), which is what gets committed under profiles/."""
import csv
import glob
import sys


def main(d, out=None, top=40):
    files = glob.glob(f"{d}/**/*kernel_stats.csv", recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_stats.csv under {d}")
    rows = list(csv.DictReader(open(files[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    lines = [f"# source: {files[0]}", f"# total kernel time: {total / 1e6:.3f} ms over {len(rows)} distinct kernels",
             f"{'calls':>7} {'total_ms':>10} {'avg_us':>10} {'pct':>6}  name"]
    for r in rows[:top]:
        lines.append(f"{int(r['Calls']):7d} {float(r['TotalDurationNs']) / 1e6:10.3f} {float(r['AverageNs']) / 1e3:10.1f} "
                     f"{float(r['Percentage']):6.2f}  {r['Name'][:150]}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
