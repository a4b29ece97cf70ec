mkdir -p gpurun_out/r04e; O=gpurun_out/r04e
python profiles/mipmap_bench.py --reps 10 --flags 0,1,2,4,8,15 > $O/mipmap_wave_ablate.log 2>&1; grep flags $O/mipmap_wave_ablate.log
for lib in product coop0; do
  if [ $lib = product ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
  python profiles/shape_bench.py --what raster --reps 10 $L --out $O/raster_$lib.json > /dev/null 2> $O/raster_$lib.log
done
for lib in product r03 notable; do
  if [ $lib = product ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
  python profiles/shape_bench.py --what interp_c --reps 20 --grads both,attr_only $L --out $O/interp_$lib.json > /dev/null 2> $O/interp_$lib.log
done
python profiles/kernel_bench.py --only interpolate_backward --check --reps 20 --mesh 250k 2>&1 | grep -v amdgpu.ids
cd tests; timeout 900 python -m pytest test_gpu_parity.py -x -q -k "interpolate or seeded or randomised or full_size or fixture or depth_fastmath or batch_at" > ../$O/tests.log 2>&1; tail -3 ../$O/tests.log; cd ..
timeout 300 python tests/fuzz_all_ops.py --cases 300 --first 86000 --wide-channels > $O/fuzz_wide.log 2>&1; tail -1 $O/fuzz_wide.log
timeout 300 python tests/fuzz_all_ops.py --cases 300 --first 87000 > $O/fuzz_all.log 2>&1; tail -1 $O/fuzz_all.log
