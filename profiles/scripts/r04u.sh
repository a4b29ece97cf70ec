#!/bin/bash
# the sampler's windowed backward kernels instantiated for double: parity (fuzzers hold f64 to 1e-10), then time against the
# direct kernel it replaces (profiles/variants/head.so = the library before this change)
mkdir -p gpurun_out/r04u
export DRTK_CAPI_POISON=1
python -m pytest tests/test_gpu_mipmap.py tests/test_gpu_f64_distance.py tests/test_gpu_textured.py -x -q > gpurun_out/r04u/tests.log 2>&1; tail -2 gpurun_out/r04u/tests.log
python tests/fuzz_mipmap.py --first 1000000 --cases 3000 > gpurun_out/r04u/fuzz_mipmap.log 2>&1; tail -2 gpurun_out/r04u/fuzz_mipmap.log
python tests/fuzz_mipmap_snapped.py --first 1010000 --cases 400 > gpurun_out/r04u/fuzz_mipmap_snapped.log 2>&1; tail -1 gpurun_out/r04u/fuzz_mipmap_snapped.log
DRTK_CAPI_GUARD=1 python tests/fuzz_mipmap.py --first 1020000 --cases 400 > gpurun_out/r04u/fuzz_mipmap_guard.log 2>&1; tail -1 gpurun_out/r04u/fuzz_mipmap_guard.log
python tests/fuzz_misaligned.py --first 1030000 --cases 150 > gpurun_out/r04u/fuzz_misaligned.log 2>&1; tail -2 gpurun_out/r04u/fuzz_misaligned.log
unset DRTK_CAPI_POISON
( python3 profiles/mipmap_bench.py --reps 3 --f64 2>&1 | grep "flags="
  python3 profiles/mipmap_bench.py --reps 3 --f64 --lib profiles/variants/head.so 2>&1 | grep "flags=" | sed 's/^/before: /'
  python3 profiles/mipmap_bench.py --reps 3 --f64 --bicubic 2>&1 | grep "flags="
  python3 profiles/mipmap_bench.py --reps 2 --f64 --bicubic --lib profiles/variants/head.so 2>&1 | grep "flags=" | sed 's/^/before: /'
  python3 profiles/mipmap_bench.py --reps 3 --f64 --channels 8 2>&1 | grep "flags="
  python3 profiles/mipmap_bench.py --reps 2 --f64 --channels 8 --lib profiles/variants/head.so 2>&1 | grep "flags=" | sed 's/^/before: /'
  python3 profiles/mipmap_bench.py --reps 5 2>&1 | grep "flags=" ) > gpurun_out/r04u/mipmap_f64.txt 2>&1
cat gpurun_out/r04u/mipmap_f64.txt
