#!/bin/bash
# Same-box A/B of rasterizer variants (profiles/variants/*.so built with drtk_amd/build.py --variant) + kernel stats + phase clocks.
export TMPDIR=/tmp
out=gpurun_out/ab_raster; mkdir -p $out; rm -f $out/*.log
if [ -z "$SKIP_PARITY" ]; then
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "raster or index or full_size or fixture or soup" > $out/pytest.log 2>&1; echo "pytest rc=$?" >> $out/pytest.log
python tests/fuzz_raster_large.py --cases 40 > $out/fuzz_raster_large.log 2>&1; echo "rc=$?" >> $out/fuzz_raster_large.log
python tests/fuzz_snapped.py --cases 100 > $out/fuzz_snapped.log 2>&1; echo "rc=$?" >> $out/fuzz_snapped.log
tail -3 $out/pytest.log; tail -2 $out/fuzz_raster_large.log; tail -2 $out/fuzz_snapped.log
fi
for rep in 1 2; do
for lib in main $VARIANTS; do
  for mesh in 100k 250k; do
    if [ $lib = main ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
    echo -n "$lib $mesh " >> $out/times.log
    python profiles/kernel_bench.py --only rasterize --reps 20 --mesh $mesh $L 2>&1 | grep "ms" >> $out/times.log
  done
done
done
cat $out/times.log
for mesh in 100k 250k; do
python profiles/raster_phases.py --mesh $mesh > $out/phases_$mesh.log 2>&1; cat $out/phases_$mesh.log
python profiles/kernel_bench.py --only rasterize --reps 10 --mesh $mesh --flags 0,1,8,64 2>&1 | grep ms > $out/flags_$mesh.log; cat $out/flags_$mesh.log
done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 profiles/kernel_bench.py --only rasterize --reps 20 > $out/stats.log 2>&1
python3 profiles/summarize_stats.py $out/stats $out/kernel_stats.txt; head -20 $out/kernel_stats.txt
