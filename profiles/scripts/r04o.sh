mkdir -p gpurun_out/r04o; O=gpurun_out/r04o
cd tests; timeout 900 python -m pytest test_gpu_parity.py -x -q -k "interpolate or seeded or randomised or fixture or full_size_view" > ../$O/tests.log 2>&1; tail -2 ../$O/tests.log; cd ..
timeout 300 python tests/fuzz_all_ops.py --cases 500 --first 600000 --wide-channels > $O/fuzz_wide.log 2>&1; tail -1 $O/fuzz_wide.log
timeout 300 python tests/fuzz_all_ops.py --cases 500 --first 610000 > $O/fuzz_all.log 2>&1; tail -1 $O/fuzz_all.log
for rep in 1 2 3; do for lib in product blocktiles; do
  if [ $lib = product ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
  python profiles/shape_bench.py --what interp_c --reps 20 --channels 8,12,16,24,32,64 --grads both,attr_only $L 2>&1 >/dev/null | grep "'C'" | awk -v l=$lib '{printf "%s %s %s %s | ", l, $2, $4, $6} END{print ""}'
done; done
