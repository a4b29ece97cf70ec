#!/bin/bash
# rasterize: element-aligned quad stores at any width -- suite, rasterizer fuzzers, guard elements, then the round's collection
mkdir -p gpurun_out/r04t
export DRTK_CAPI_POISON=1
python -m pytest tests -m gpu -x -q > gpurun_out/r04t/tests.log 2>&1; tail -2 gpurun_out/r04t/tests.log
python tests/fuzz_all_ops.py --first 950000 --cases 1500 > gpurun_out/r04t/fuzz_all_ops.log 2>&1; tail -1 gpurun_out/r04t/fuzz_all_ops.log
python tests/fuzz_raster_large.py --first 960000 --cases 300 > gpurun_out/r04t/fuzz_raster_large.log 2>&1; tail -1 gpurun_out/r04t/fuzz_raster_large.log
python tests/fuzz_snapped.py --first 970000 --cases 500 > gpurun_out/r04t/fuzz_snapped.log 2>&1; tail -1 gpurun_out/r04t/fuzz_snapped.log
DRTK_CAPI_GUARD=3 python tests/fuzz_all_ops.py --first 980000 --cases 500 > gpurun_out/r04t/fuzz_guard3.log 2>&1; tail -1 gpurun_out/r04t/fuzz_guard3.log
python tests/fuzz_misaligned.py --first 990000 --cases 200 > gpurun_out/r04t/fuzz_misaligned.log 2>&1; tail -2 gpurun_out/r04t/fuzz_misaligned.log
python tests/fuzz_large_scenes.py --first 995000 --cases 20 > gpurun_out/r04t/fuzz_large.log 2>&1; tail -1 gpurun_out/r04t/fuzz_large.log
unset DRTK_CAPI_POISON
python profiles/host_time.py --out gpurun_out/r04t/host_time.txt > gpurun_out/r04t/host_time.log 2>&1; head -3 gpurun_out/r04t/host_time.txt
bash profiles/scripts/collect_r04.sh > gpurun_out/r04t/collect.log 2>&1; tail -5 gpurun_out/r04t/collect.log | cut -c1-300
