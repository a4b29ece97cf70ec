mkdir -p gpurun_out/r04m; O=gpurun_out/r04m
cd tests; timeout 900 python -m pytest test_gpu_mipmap.py -x -q > ../$O/tests.log 2>&1; tail -2 ../$O/tests.log; cd ..
timeout 400 python tests/fuzz_mipmap.py --cases 1200 --first 540000 > $O/fuzz_mipmap.log 2>&1; tail -1 $O/fuzz_mipmap.log
timeout 300 python tests/fuzz_mipmap_snapped.py --cases 200 --first 550000 > $O/fuzz_mipmap_snapped.log 2>&1; tail -1 $O/fuzz_mipmap_snapped.log
for rep in 1 2; do for lib in product head; do
  if [ $lib = product ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
  python profiles/mipmap_bench.py --reps 5 --channels 3 --bicubic $L 2>&1 | grep flags= | sed "s/^/$lib /"
  python profiles/kernel_bench.py --only mipmap_fwd_bicubic --reps 5 $L 2>&1 | grep ms | sed "s/^/$lib /"
done; done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
