# Re-measures the table of profiles/NOTES.md section 5 of the other BASELINE.json configurations (parity cases, not bench lines):
# one bench.py run each, 10 timed steps (200 for the launch-bound configs[1]), CPU-baseline leg skipped.  One JSON line per row in gpurun_out/configs/.
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/configs
rm -rf $OUT; mkdir -p $OUT
run() { name=$1; shift; timeout 400 python3 bench.py --steps ${STEPS:-10} --warmup 3 --cpu-sample-views 0 "$@" 2> $OUT/$name.err | grep "^{" > $OUT/$name.json; }
STEPS=200 run config2_10k_512_v4_c3 --config 2   # a 0.4 ms step: ten of them are shorter than one scheduler hiccup
run config3_100k_2048_v8_c16      --config 3
run config4_250k_2048_v8_c16      --config 4
run config5_textured_1M_4096_v2   --config 5
run geom5_1M_4096_v8_c16          --mesh 1M --res 4096 --views 8 --channels 16
run geom5_1M_4096_v2_c3           --mesh 1M --res 4096 --views 2 --channels 3
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/configs/*.json")):
    txt = open(f).read().strip()
    if not txt:
        print(f, "EMPTY"); continue
    b = json.loads(txt.splitlines()[-1]); p = b["path_roofline"]; o = p["ops_ms"]
    g = b.get("graph_step") or {}
    print(f.split("/")[-1][:-5], f"{b['value']:.0f} Mpix/s  {b['ms_per_step']:.3f} ms/step (graph replay {g.get('ms_per_step')})  t_ops {p['t_ops_ms']:.3f}  frac_ops {p['frac_ops']:.3f} | "
          + " / ".join(f"{k} {v:.3f}" for k, v in o.items()))
PY
