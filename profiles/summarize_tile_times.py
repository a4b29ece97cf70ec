"""Text summary of the ablation build's per-tile dumps of the lean sampler backward (profiles/mipmap_bench.py
--tile-times / --tile-phases / --leftover-dump):  python3 profiles/summarize_tile_times.py times.npz phases.npz leftover.npz"""
import sys

import numpy as np


def times(path):
    d = np.load(path)
    r = d["rounds"].astype(int)
    t0, t1 = d["t0"].astype(np.int64), d["t1"].astype(np.int64)
    base = t0.min()
    t0, t1 = t0 - base, t1 - base
    dur = (t1 - t0) * 0.01
    print(f"tile timeline ({path}): {len(r)} tiles with upstream gradient, kernel span {t1.max() * 0.01:.1f} us (10 ns clock)")
    for k in sorted(set(r.tolist())):
        m = r == k
        print(f"  further rounds = {k:2d}: {m.sum():6d} tiles, mean life {dur[m].mean():7.2f} us, p90 {np.percentile(dur[m], 90):7.2f}, sum {dur[m].sum() / 1000:8.2f} ms")
    print(f"  sum of tile lives {dur.sum() / 1000:.1f} ms = {dur.sum() / 1000 / 1024:.3f} ms on 1024 tile slots")
    print("  finish times (us), percentiles 50 / 90 / 99 / 99.9 / 100:", [round(float(x) * 0.01, 1) for x in np.percentile(t1, [50, 90, 99, 99.9, 100])])
    edges = np.linspace(0, t1.max(), 21)[1:-1]
    print("  tiles in flight at 5 % ... 95 % of the span:", [int(((t0 < e) & (t1 > e)).sum()) for e in edges])


def phases(path):
    d = np.load(path)
    r = d["rounds"]
    rounds = r & 0xFFFF
    a, b, c, e = (r >> 16) * 0.01, (d["t0"] & 0xFFFF) * 0.01, (d["t0"] >> 16) * 0.01, d["t1"] * 0.01
    print(f"tile phases ({path}), us since the tile's start:")
    for k in (0, 1, 2, 3, 5):
        m = rounds == k
        if m.sum():
            print(f"  further rounds = {k}: {m.sum():6d} tiles: inputs there {a[m].mean():.2f}, windows placed +{(b[m] - a[m]).mean():.2f}, "
                  f"taps done +{(c[m] - b[m]).mean():.2f}, end +{(e[m] - c[m]).mean():.2f}, life {e[m].mean():.2f}")


def leftover(path):
    d = np.load(path)
    view, tile, lev, x, y = (d[k] for k in ("view", "tile", "level", "x", "y"))
    key = view.astype(np.int64) << 24 | tile
    u, inv, cnt = np.unique(key, return_inverse=True, return_counts=True)
    print(f"left-over pairs ({path}): {len(tile)} (tap, level) pairs end in global memory after the last round, in {len(u)} tiles; "
          f"pairs per tile p50 / p90 / max {np.percentile(cnt, 50):.0f} / {np.percentile(cnt, 90):.0f} / {cnt.max()}" if len(tile) else "left-over pairs: none")
    if len(tile):
        print("  by level:", np.bincount(lev).tolist())
        tiles_x = int(d["W"]) // 16
        ty = (u & 0xFFFFFF) // tiles_x
        print(f"  by tile row (16 bins over {int(d['H']) // 16} rows):", np.histogram(ty, bins=16, range=(0, int(d["H"]) // 16), weights=cnt)[0].astype(int).tolist())


for p in sys.argv[1:]:
    z = np.load(p)
    if "thread" in z.files:
        leftover(p)
    elif int((z["rounds"] >> 16).max()) > 0:
        phases(p)
    else:
        times(p)
