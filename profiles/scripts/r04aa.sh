#!/bin/bash
# screen_space_uv_derivative with four pixels per lane: parity, then time against the one-pixel kernel (variant uv1px.so)
mkdir -p gpurun_out/r04aa
export DRTK_CAPI_POISON=1
python -m pytest tests/test_gpu_mipmap.py tests/test_gpu_textured.py tests/test_gpu_bench_contract.py -x -q > gpurun_out/r04aa/tests.log 2>&1; tail -1 gpurun_out/r04aa/tests.log
python tests/fuzz_next_ops.py --first 1400000 --cases 3000 > gpurun_out/r04aa/fuzz_next_ops.log 2>&1; grep -c "^FAIL" gpurun_out/r04aa/fuzz_next_ops.log; tail -1 gpurun_out/r04aa/fuzz_next_ops.log
DRTK_CAPI_GUARD=1 python tests/fuzz_next_ops.py --first 1410000 --cases 300 > gpurun_out/r04aa/fuzz_guard.log 2>&1; tail -1 gpurun_out/r04aa/fuzz_guard.log
unset DRTK_CAPI_POISON
for i in 1 2; do
python3 profiles/mipmap_bench.py --reps 10 --uv 2>&1 | grep "screen_space"
python3 profiles/mipmap_bench.py --reps 10 --uv --lib profiles/variants/uv1px.so 2>&1 | grep "screen_space" | sed 's/^/one pixel per lane: /'
done
