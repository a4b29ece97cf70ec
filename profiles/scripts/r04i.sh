# the wide nets on the round's kernels + the whole GPU suite
mkdir -p gpurun_out/r04i; O=gpurun_out/r04i
cd tests; timeout 1800 python -m pytest . -m gpu -q > ../$O/tests.log 2>&1; tail -3 ../$O/tests.log; cd ..
run() { name=$1; shift; timeout 900 python "$@" > $O/$name.log 2>&1; echo "$name: $(tail -1 $O/$name.log)"; }
run fuzz_all_ops        tests/fuzz_all_ops.py --cases 3000 --first 400000
run fuzz_all_ops_wide   tests/fuzz_all_ops.py --cases 2000 --first 410000 --wide-channels
run fuzz_raster_large   tests/fuzz_raster_large.py --cases 700 --first 420000
run fuzz_snapped        tests/fuzz_snapped.py --cases 1500 --first 430000
run fuzz_mipmap         tests/fuzz_mipmap.py --cases 3000 --first 440000
run fuzz_mipmap_snapped tests/fuzz_mipmap_snapped.py --cases 600 --first 450000
run fuzz_next_ops       tests/fuzz_next_ops.py --cases 1500 --first 460000
run fuzz_python_api     tests/fuzz_python_api.py --cases 600 --first 470000
run fuzz_large_scenes   tests/fuzz_large_scenes.py --cases 40 --first 480000
run fuzz_misaligned     tests/fuzz_misaligned.py
