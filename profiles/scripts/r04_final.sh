# the round's last pass: the GPU suite, a fuzz pass on the final kernels, then the committed evidence (collect_r04.sh)
mkdir -p gpurun_out/r04z; O=gpurun_out/r04z
cd tests; timeout 1800 python -m pytest . -m gpu -q > ../$O/tests.log 2>&1; tail -3 ../$O/tests.log; cd ..
run() { name=$1; shift; timeout 900 python "$@" > $O/$name.log 2>&1; echo "$name: $(tail -1 $O/$name.log)"; }
run fuzz_all_ops        tests/fuzz_all_ops.py --cases 2000 --first 700000
run fuzz_all_ops_wide   tests/fuzz_all_ops.py --cases 1500 --first 710000 --wide-channels
run fuzz_raster_large   tests/fuzz_raster_large.py --cases 400 --first 720000
run fuzz_snapped        tests/fuzz_snapped.py --cases 800 --first 730000
run fuzz_mipmap         tests/fuzz_mipmap.py --cases 2500 --first 740000
run fuzz_mipmap_snapped tests/fuzz_mipmap_snapped.py --cases 400 --first 750000
run fuzz_next_ops       tests/fuzz_next_ops.py --cases 1500 --first 760000
run fuzz_python_api     tests/fuzz_python_api.py --cases 400 --first 770000
run fuzz_large_scenes   tests/fuzz_large_scenes.py --cases 30 --first 780000
bash profiles/scripts/collect_r04.sh > $O/collect.log 2>&1; tail -6 $O/collect.log
