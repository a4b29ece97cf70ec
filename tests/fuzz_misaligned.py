"""The all-ops and mipmap fuzzers with EVERY input placed one element into a flat buffer (contiguous, but the pointer
is only element-aligned): 16-byte vector paths must not be taken, results must not change.
usage: python tests/fuzz_misaligned.py [--first S] [--cases K]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fuzz_all_ops as FA
import fuzz_mipmap as FM

ap = argparse.ArgumentParser(); ap.add_argument("--first", type=int, default=0); ap.add_argument("--cases", type=int, default=100)
a = ap.parse_args(); rc = 0
for name, mod in (("all_ops", FA), ("mipmap", FM)):
    bad = 0
    for seed in range(a.first, a.first + a.cases):
        c = mod.make_case(seed)
        try:
            mod.run_case(c, place=FA.misaligned)
        except Exception as e:
            bad += 1
            print(f"FAIL {name} seed {seed}: {mod.describe(c)}: {type(e).__name__}: {str(e)[:200]}", flush=True)
    print(f"{name} (misaligned inputs): {a.cases - bad}/{a.cases} cases passed", flush=True)
    rc |= bad > 0
sys.exit(rc)
