#!/usr/bin/env python3
"""Condense a rocprofv3 --pmc run (counter_collection.csv) into per-kernel mean counter values."""
import csv
import glob
import sys
from collections import defaultdict


def main(d, out=None, match=""):
    files = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"]
            if match and match not in name:
                continue
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = []
    for name, ctrs in acc.items():
        lines.append(name[:140])
        for c, vals in sorted(ctrs.items()):
            lines.append(f"    {c:28s} mean {sum(vals) / len(vals):16.1f}  (n={len(vals)})")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, sys.argv[3] if len(sys.argv) > 3 else "")
