mkdir -p gpurun_out/r04k; O=gpurun_out/r04k
cd tests; timeout 900 python -m pytest test_gpu_parity.py -x -q -k "cooperative" > ../$O/tests.log 2>&1; tail -2 ../$O/tests.log; cd ..
for rep in 1 2; do for lib in lean nolean; do
  python profiles/mipmap_bench.py --reps 10 --lib profiles/variants/$lib.so 2>&1 | grep flags= | sed "s/^/$lib /"
  python profiles/kernel_bench.py --only mipmap_bwd --reps 10 --lib profiles/variants/$lib.so 2>&1 | grep ms | sed "s/^/$lib 1tx /"
done; done
python profiles/mipmap_bench.py --dump $O/lean.pt --lib profiles/variants/lean.so > /dev/null 2>&1
python profiles/mipmap_bench.py --dump $O/nolean.pt --lib profiles/variants/nolean.so > /dev/null 2>&1
python profiles/mipmap_bench.py --compare $O/lean.pt $O/nolean.pt
for c in 8 16; do python profiles/mipmap_bench.py --reps 3 --channels $c --flags 0,512 2>&1 | grep flags=; done
