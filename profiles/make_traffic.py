#!/usr/bin/env python3
"""Builds profiles/rNN/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over
profiles/kernel_bench.py:  hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) KiB per launch -- FETCH_SIZE reports
half of a wide coalesced read on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section).
    python3 profiles/make_traffic.py <dir with FETCH_SIZE/ and WRITE_SIZE/ runs> <out.json>"""
import csv
import glob
import json
import sys
from collections import defaultdict


def mean_counter(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def short(name):
    name = name.replace("void ", "").replace("drtk_amd::(anonymous namespace)::", "")
    return name.split("(")[0]


def main(d, out):
    fetch = mean_counter(f"{d}/FETCH_SIZE", "FETCH_SIZE")
    write = mean_counter(f"{d}/WRITE_SIZE", "WRITE_SIZE")
    kernels = {}
    for k in fetch:
        if "drtk_amd" not in k:
            continue
        f, w = fetch[k], write.get(k, 0.0)
        kernels[short(k)] = {"fetch_kb": round(f, 1), "write_kb": round(w, 1), "hbm_bytes": int((2 * f + w) * 1024)}
    import hashlib
    import os

    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "drtk_amd", "csrc")
    sources = {f: hashlib.sha256(open(os.path.join(csrc, f), "rb").read()).hexdigest()
               for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".hpp"))}
    doc = {
        # what the counters were taken on: bench.py compares these with the tree it runs from (roofline.traffic_stale)
        "sources": sources,
        "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on `python3 profiles/kernel_bench.py --reps 2`, "
                "bench workload (8 views, 100k tris, 2048^2, C=16); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) KiB per "
                "MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide coalesced read on gfx950)",
        "kernels": kernels,
    }
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
