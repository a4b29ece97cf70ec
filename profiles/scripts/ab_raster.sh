#!/bin/bash
# Same-box A/B of rasterizer variants (profiles/variants/*.so built with drtk_amd/build.py --variant): parity first, then times.
out=gpurun_out/ab_raster; mkdir -p $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "raster or index or full_size or fixture or soup" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
python tests/fuzz_raster_large.py --cases 40 > $out/fuzz_raster_large.log 2>&1; echo "rc=$?" >> $out/fuzz_raster_large.log
python tests/fuzz_snapped.py --cases 150 > $out/fuzz_snapped.log 2>&1; echo "rc=$?" >> $out/fuzz_snapped.log
for rep in 1 2; do
for lib in main legacy w4 w8 noslp; do
  for mesh in 100k 250k; do
    if [ $lib = main ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
    echo "== $lib $mesh" >> $out/times.log
    python profiles/kernel_bench.py --only rasterize --reps 20 --mesh $mesh $L >> $out/times.log 2>&1
  done
done
done
tail -3 $out/pytest.log; tail -2 $out/fuzz_raster_large.log; tail -2 $out/fuzz_snapped.log; cat $out/times.log
