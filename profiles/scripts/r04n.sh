mkdir -p gpurun_out/r04n; O=gpurun_out/r04n
timeout 300 python tests/fuzz_all_ops.py --cases 400 --first 570000 --wide-channels > $O/fuzz_wide.log 2>&1; tail -1 $O/fuzz_wide.log
for rep in 1 2 3; do for lib in product head; do
  if [ $lib = product ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
  python profiles/shape_bench.py --what interp_c --reps 20 --channels 16,32,24 --grads both,attr_only $L 2>&1 >/dev/null | grep "'C'" | awk -v l=$lib '{printf "%s %s %s %s | ", l, $2, $4, $6} END{print ""}'
done; done
