#!/bin/bash
# the suite on the latest build (f64 sampler windows, bench tie-break), then f64 interpolate forward with / without the prefetching loop
mkdir -p gpurun_out/r04v
export DRTK_CAPI_POISON=1
python -m pytest tests -m gpu -x -q > gpurun_out/r04v/tests.log 2>&1; tail -2 gpurun_out/r04v/tests.log
unset DRTK_CAPI_POISON
for i in 1 2; do
python3 profiles/shape_bench.py --what f64 --reps 10 2> gpurun_out/r04v/f64_product_$i.log > /dev/null; grep "f64 2048" gpurun_out/r04v/f64_product_$i.log | cut -c1-200
python3 profiles/shape_bench.py --what f64 --reps 10 --lib profiles/variants/f64_noprefetch.so 2> gpurun_out/r04v/f64_noprefetch_$i.log > /dev/null; grep "f64 2048" gpurun_out/r04v/f64_noprefetch_$i.log | sed 's/^/no prefetch: /' | cut -c1-200
done
python bench.py > gpurun_out/r04v/bench.json 2> gpurun_out/r04v/bench.err; python -c "
import json; d=json.loads([l for l in open('gpurun_out/r04v/bench.json') if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['path_roofline']['t_ops_ms'])"
