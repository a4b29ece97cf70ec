#!/bin/bash
# the G^-1 A form of screen_space_uv_derivative, the zero-fill folded into edge_dots, rasterize without an autograd node:
# the GPU suite, the fuzzers that cover them, the accuracy distribution, then config 2 and the headline
mkdir -p gpurun_out/r04q
export DRTK_CAPI_POISON=1
python -m pytest tests -m gpu -x -q > gpurun_out/r04q/tests.log 2>&1; tail -2 gpurun_out/r04q/tests.log
for s in 760460 450324 460768; do python tests/fuzz_next_ops.py --first $s --cases 1 >> gpurun_out/r04q/seeds.log 2>&1; done; grep -v amdgpu gpurun_out/r04q/seeds.log | tail -6
python tests/fuzz_next_ops.py --first 800000 --cases 3000 > gpurun_out/r04q/fuzz_next_ops.log 2>&1; tail -2 gpurun_out/r04q/fuzz_next_ops.log
python tests/diag_uv_derivative_accuracy.py --first 10000 --cases 1500 > gpurun_out/r04q/uv_accuracy.log 2>&1; tail -3 gpurun_out/r04q/uv_accuracy.log
python tests/fuzz_all_ops.py --first 810000 --cases 1000 > gpurun_out/r04q/fuzz_all_ops.log 2>&1; tail -1 gpurun_out/r04q/fuzz_all_ops.log
python tests/fuzz_python_api.py --first 820000 --cases 300 > gpurun_out/r04q/fuzz_python_api.log 2>&1; tail -1 gpurun_out/r04q/fuzz_python_api.log
unset DRTK_CAPI_POISON
python profiles/host_time.py --out gpurun_out/r04q/host_time.txt > gpurun_out/r04q/host_time.log 2>&1; head -12 gpurun_out/r04q/host_time.txt
for i in 1 2 3; do python bench.py --config 2 --steps 200 --warmup 20 --cpu-sample-views 0 > gpurun_out/r04q/config2_$i.json 2> gpurun_out/r04q/config2_$i.err; done
python bench.py > gpurun_out/r04q/bench.json 2> gpurun_out/r04q/bench.err
python - <<'P'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04q/config2_*.json'))+['gpurun_out/r04q/bench.json']:
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, d['value'], d['ms_per_step'], d.get('ms_per_step_median_hipevent'), d['graph_step']['ms_per_step'] if d.get('graph_step') else None, d['extensions']['operators_only']['ms_per_step'], d['path_roofline']['t_ops_ms'])
    except Exception as e: print(f, 'ERR', e)
P
