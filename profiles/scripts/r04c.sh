mkdir -p gpurun_out/r04c; O=gpurun_out/r04c
cd tests; timeout 900 python -m pytest test_gpu_parity.py -x -q -k "fixture or seeded or soup or large_random or batch_at_bench or full_size or randomised or many_small or zero_sized or non_finite or odd_element" > ../$O/tests.log 2>&1; tail -3 ../$O/tests.log; cd ..
timeout 500 python tests/fuzz_raster_large.py --cases 300 --first 80000 > $O/fuzz_raster_large.log 2>&1; tail -2 $O/fuzz_raster_large.log
timeout 300 python tests/fuzz_snapped.py --cases 300 --first 81000 > $O/fuzz_snapped.log 2>&1; tail -2 $O/fuzz_snapped.log
for lib in product coop0 coop256 coop1024; do
  if [ $lib = product ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
  python profiles/shape_bench.py --what raster --reps 10 $L --out $O/raster_$lib.json > /dev/null 2> $O/raster_$lib.log
done
for F in 0 1 32 8; do python profiles/shape_bench.py --what interp_c --channels 8,12,16,24 --grads attr_only,both --reps 10 --flags $F --out $O/interp_flags$F.json > /dev/null 2> $O/interp_flags$F.log; done
python profiles/shape_bench.py --what interp_c --channels 16,32 --grads both --reps 20 --lib profiles/variants/qa2.so --out $O/interp_qa2.json > /dev/null 2> $O/interp_qa2.log
python profiles/shape_bench.py --what interp_c --channels 16,32 --grads both --reps 20 --out $O/interp_qa1.json > /dev/null 2> $O/interp_qa1.log
python profiles/shape_bench.py --what interp_c --channels 16,32 --grads both --reps 20 --lib profiles/variants/r03.so --out $O/interp_r03.json > /dev/null 2> $O/interp_r03.log
