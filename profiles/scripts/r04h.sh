mkdir -p gpurun_out/r04h; O=gpurun_out/r04h
cd tests; timeout 900 python -m pytest test_gpu_mipmap.py test_gpu_textured.py test_gpu_f64_distance.py -x -q > ../$O/tests.log 2>&1; tail -3 ../$O/tests.log; cd ..
timeout 400 python tests/fuzz_mipmap.py --cases 800 --first 91000 > $O/fuzz_mipmap.log 2>&1; tail -2 $O/fuzz_mipmap.log
timeout 300 python tests/fuzz_mipmap_snapped.py --cases 200 --first 92000 > $O/fuzz_mipmap_snapped.log 2>&1; tail -2 $O/fuzz_mipmap_snapped.log
for rep in 1 2; do for lib in product head; do
  if [ $lib = product ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
  python profiles/mipmap_bench.py --reps 10 $L > $O/mipmap_$lib.log 2>&1; echo "$lib $(tail -1 $O/mipmap_$lib.log)"
  python profiles/kernel_bench.py --only mipmap_bwd --reps 10 $L 2>&1 | grep ms | sed "s/^/$lib 1tx /"
  python profiles/kernel_bench.py --only mipmap_bwd --reps 10 --uvscale 4 $L 2>&1 | grep ms | sed "s/^/$lib 4tx /"
done; done
python profiles/mipmap_bench.py --reps 10 --flags 0,15,47,32 > $O/mipmap_tiled_ablate.log 2>&1; grep flags $O/mipmap_tiled_ablate.log
