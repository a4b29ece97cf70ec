#!/bin/bash
# interpolate forward on four-pixel lanes at any width: parity tests, fuzzers (plain / misaligned / guard elements), W = 2046 timings
mkdir -p gpurun_out/r04s
export DRTK_CAPI_POISON=1
python -m pytest tests -m gpu -x -q > gpurun_out/r04s/tests.log 2>&1; tail -2 gpurun_out/r04s/tests.log
python tests/fuzz_all_ops.py --first 900000 --cases 2000 > gpurun_out/r04s/fuzz_all_ops.log 2>&1; tail -1 gpurun_out/r04s/fuzz_all_ops.log
python tests/fuzz_all_ops.py --first 910000 --cases 600 --wide-channels > gpurun_out/r04s/fuzz_wide.log 2>&1; tail -1 gpurun_out/r04s/fuzz_wide.log
python tests/fuzz_misaligned.py --first 920000 --cases 300 > gpurun_out/r04s/fuzz_misaligned.log 2>&1; tail -2 gpurun_out/r04s/fuzz_misaligned.log
DRTK_CAPI_GUARD=1 python tests/fuzz_all_ops.py --first 930000 --cases 500 > gpurun_out/r04s/fuzz_guard1.log 2>&1; tail -1 gpurun_out/r04s/fuzz_guard1.log
python tests/fuzz_python_api.py --first 940000 --cases 300 > gpurun_out/r04s/fuzz_python_api.log 2>&1; tail -1 gpurun_out/r04s/fuzz_python_api.log
unset DRTK_CAPI_POISON
python3 profiles/shape_bench.py --what f64 --reps 10 > gpurun_out/r04s/shape_f64.json 2> gpurun_out/r04s/shape_f64.log; tail -3 gpurun_out/r04s/shape_f64.log
python profiles/host_time.py --out gpurun_out/r04s/host_time.txt > gpurun_out/r04s/host_time.log 2>&1; head -3 gpurun_out/r04s/host_time.txt
python bench.py > gpurun_out/r04s/bench.json 2> gpurun_out/r04s/bench.err; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r04s/bench.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['path_roofline']['t_ops_ms'])"
