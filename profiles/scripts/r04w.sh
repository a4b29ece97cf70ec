#!/bin/bash
# the round's closing campaign on the final build: suite, smoke(), every fuzzer on fresh seeds (uv-derivative factor now 2)
mkdir -p gpurun_out/r04w
export DRTK_CAPI_POISON=1
python -m pytest tests -m gpu -x -q > gpurun_out/r04w/tests.log 2>&1; tail -2 gpurun_out/r04w/tests.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r04w/smoke.log 2>&1; tail -1 gpurun_out/r04w/smoke.log
run() { name=$1; shift; python tests/$name.py "$@" > gpurun_out/r04w/$name$SUF.log 2>&1; echo "$name$SUF: $(grep -c '^FAIL' gpurun_out/r04w/$name$SUF.log) failures; $(tail -1 gpurun_out/r04w/$name$SUF.log | cut -c1-150)"; }
run fuzz_next_ops --first 1100000 --cases 4000
run fuzz_all_ops --first 1110000 --cases 2000
SUF=_wide run fuzz_all_ops --first 1120000 --cases 600 --wide-channels
run fuzz_mipmap --first 1130000 --cases 2000
run fuzz_mipmap_snapped --first 1140000 --cases 300
run fuzz_raster_large --first 1150000 --cases 200
run fuzz_snapped --first 1160000 --cases 400
run fuzz_python_api --first 1170000 --cases 300
run fuzz_large_scenes --first 1180000 --cases 20
run fuzz_misaligned --first 1190000 --cases 200
SUF=_guard2 DRTK_CAPI_GUARD=2 run fuzz_next_ops --first 1200000 --cases 300
