mkdir -p gpurun_out/r04l; O=gpurun_out/r04l
cd tests; timeout 900 python -m pytest test_gpu_mipmap.py test_gpu_textured.py test_gpu_f64_distance.py -x -q > ../$O/tests.log 2>&1; tail -3 ../$O/tests.log; cd ..
timeout 600 python tests/fuzz_mipmap.py --cases 2000 --first 520000 > $O/fuzz_mipmap.log 2>&1; tail -2 $O/fuzz_mipmap.log
timeout 400 python tests/fuzz_mipmap_snapped.py --cases 400 --first 530000 > $O/fuzz_mipmap_snapped.log 2>&1; tail -2 $O/fuzz_mipmap_snapped.log
timeout 200 python tests/fuzz_misaligned.py > $O/fuzz_mis.log 2>&1; tail -1 $O/fuzz_mis.log
python profiles/mipmap_bench.py --reps 3 --channels 3 --bicubic --flags 0,512 2>&1 | grep flags=
python profiles/mipmap_bench.py --reps 3 --channels 8 --bicubic 2>&1 | grep flags=
python profiles/kernel_bench.py --only mipmap_bwd_bicubic,mipmap_bwd --reps 3 2>&1 | grep ms
