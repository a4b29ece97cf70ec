mkdir -p gpurun_out/r04f; O=gpurun_out/r04f
cd tests; timeout 1500 python -m pytest . -m gpu -x -q > ../$O/tests.log 2>&1; tail -3 ../$O/tests.log; cd ..
timeout 300 python tests/fuzz_all_ops.py --cases 400 --first 88000 --wide-channels > $O/fuzz_wide.log 2>&1; tail -1 $O/fuzz_wide.log
for lib in product wideonly r03; do
  if [ $lib = product ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
  python profiles/shape_bench.py --what interp_c --reps 20 --channels 8,12,16,20,24,32,40,64 --grads both,attr_only $L --out $O/interp_$lib.json > /dev/null 2> $O/interp_$lib.log
done
python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err; cat $O/bench_n1.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['path_roofline']['t_ops_ms'], d['path_roofline']['frac_ops'], d['path_roofline']['ops_ms'])"
python bench.py --config 5 --steps 10 --warmup 3 > $O/bench_c5.json 2> $O/bench_c5.err; cat $O/bench_c5.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['path_roofline']['t_ops_ms'], d['path_roofline']['frac_ops'], d['path_roofline']['ops_ms'])"
