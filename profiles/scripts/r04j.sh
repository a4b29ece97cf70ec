mkdir -p gpurun_out/r04j; O=gpurun_out/r04j
python tests/diag_one_seed_next_ops.py 460768 product 2>&1 | grep -v amdgpu.ids
python tests/diag_one_seed_next_ops.py 460768 profiles/variants/r03.so 2>&1 | grep -v amdgpu.ids
timeout 200 python tests/fuzz_large_scenes.py --cases 3 --first 480019 2>&1 | tail -2
cd tests; timeout 1500 python -m pytest test_gpu_parity.py test_gpu_textured.py test_gpu_f64_distance.py -x -q > ../$O/tests.log 2>&1; tail -3 ../$O/tests.log; cd ..
timeout 600 python tests/fuzz_all_ops.py --cases 1500 --first 500000 > $O/fuzz_all.log 2>&1; tail -1 $O/fuzz_all.log
timeout 300 python tests/fuzz_python_api.py --cases 300 --first 510000 > $O/fuzz_api.log 2>&1; tail -1 $O/fuzz_api.log
timeout 300 python tests/fuzz_misaligned.py > $O/fuzz_mis.log 2>&1; tail -2 $O/fuzz_mis.log
for rep in 1 2; do for lib in product head; do
  if [ $lib = product ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
  python profiles/kernel_bench.py --only edge_grad_backward_fused,edge_grad_backward,interpolate --reps 20 $L 2>&1 | grep ms | tr '\n' ' ' | sed "s/^/$lib C16: /"; echo
  python profiles/kernel_bench.py --only edge_grad_backward_fused,interpolate --reps 20 --channels 3 --mesh 1M --res 4096 --views 2 $L 2>&1 | grep ms | tr '\n' ' ' | sed "s/^/$lib 1M C3: /"; echo
done; done
