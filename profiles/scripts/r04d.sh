mkdir -p gpurun_out/r04d; O=gpurun_out/r04d
cd tests; timeout 1200 python -m pytest test_gpu_parity.py test_gpu_mipmap.py test_gpu_textured.py test_gpu_f64_distance.py -q > ../$O/tests.log 2>&1; tail -3 ../$O/tests.log; cd ..
timeout 300 python tests/fuzz_mipmap.py --cases 600 --first 90000 > $O/fuzz_mipmap.log 2>&1; tail -2 $O/fuzz_mipmap.log
timeout 300 python tests/fuzz_raster_large.py --cases 200 --first 83000 > $O/fuzz_raster_large.log 2>&1; tail -1 $O/fuzz_raster_large.log
timeout 200 python tests/fuzz_snapped.py --cases 300 --first 84000 > $O/fuzz_snapped.log 2>&1; tail -1 $O/fuzz_snapped.log
timeout 300 python tests/fuzz_all_ops.py --cases 300 --first 85000 --wide-channels > $O/fuzz_wide.log 2>&1; tail -1 $O/fuzz_wide.log
for lib in product coop0; do
  if [ $lib = product ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
  python profiles/shape_bench.py --what raster --reps 10 $L --out $O/raster_$lib.json > /dev/null 2> $O/raster_$lib.log
done
for lib in product r03 nosplit; do
  if [ $lib = product ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
  python profiles/shape_bench.py --what interp_c --reps 20 --grads both,attr_only $L --out $O/interp_$lib.json > /dev/null 2> $O/interp_$lib.log
done
for lib in product tiled; do
  if [ $lib = product ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
  python profiles/mipmap_bench.py --reps 10 $L > $O/mipmap_$lib.log 2>&1; tail -1 $O/mipmap_$lib.log
  python profiles/kernel_bench.py --only mipmap_bwd --reps 10 $L 2>&1 | grep ms > $O/kb_mipmap_$lib.log; cat $O/kb_mipmap_$lib.log
  python profiles/kernel_bench.py --only mipmap_bwd --reps 10 --uvscale 4 $L 2>&1 | grep ms > $O/kb4_mipmap_$lib.log; cat $O/kb4_mipmap_$lib.log
done
