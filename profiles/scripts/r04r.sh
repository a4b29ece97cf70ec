#!/bin/bash
# any-width fused edge route + render (element-aligned 16-byte accesses), guard-element mode of the binding:
# the GPU suite, the fuzzers (plain, misaligned inputs), W = 2046 timings, config 2 and the headline
mkdir -p gpurun_out/r04r
export DRTK_CAPI_POISON=1
python -m pytest tests -m gpu -x -q > gpurun_out/r04r/tests.log 2>&1; tail -2 gpurun_out/r04r/tests.log
python tests/fuzz_all_ops.py --first 830000 --cases 2000 > gpurun_out/r04r/fuzz_all_ops.log 2>&1; tail -1 gpurun_out/r04r/fuzz_all_ops.log
python tests/fuzz_misaligned.py --first 840000 --cases 400 > gpurun_out/r04r/fuzz_misaligned.log 2>&1; tail -2 gpurun_out/r04r/fuzz_misaligned.log
DRTK_CAPI_GUARD=2 python tests/fuzz_all_ops.py --first 850000 --cases 600 > gpurun_out/r04r/fuzz_guard2.log 2>&1; tail -1 gpurun_out/r04r/fuzz_guard2.log
python tests/fuzz_python_api.py --first 860000 --cases 300 > gpurun_out/r04r/fuzz_python_api.log 2>&1; tail -1 gpurun_out/r04r/fuzz_python_api.log
python tests/fuzz_large_scenes.py --first 870000 --cases 20 > gpurun_out/r04r/fuzz_large.log 2>&1; tail -1 gpurun_out/r04r/fuzz_large.log
python tests/fuzz_snapped.py --first 880000 --cases 300 > gpurun_out/r04r/fuzz_snapped.log 2>&1; tail -1 gpurun_out/r04r/fuzz_snapped.log
unset DRTK_CAPI_POISON
python3 profiles/shape_bench.py --what f64 --reps 10 > gpurun_out/r04r/shape_f64.json 2> gpurun_out/r04r/shape_f64.log; tail -3 gpurun_out/r04r/shape_f64.log
for i in 1 2; do python bench.py --config 2 --steps 200 --warmup 20 --cpu-sample-views 0 > gpurun_out/r04r/config2_$i.json 2> gpurun_out/r04r/config2_$i.err; done
python bench.py > gpurun_out/r04r/bench.json 2> gpurun_out/r04r/bench.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04r/config2_*.json'))+['gpurun_out/r04r/bench.json']:
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_median_hipevent'), d['graph_step']['ms_per_step'] if d.get('graph_step') else None, d['extensions']['operators_only']['ms_per_step'], d['path_roofline']['t_ops_ms'])
    except Exception as e: print(f, 'ERR', e)
P
