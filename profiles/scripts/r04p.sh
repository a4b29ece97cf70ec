#!/bin/bash
# seed 760460 of fuzz_next_ops on the product and on round 3's library; host-time profile of BASELINE configs[1]
mkdir -p gpurun_out/r04p
export DRTK_CAPI_POISON=1
python tests/diag_one_seed_next_ops.py 760460 product > gpurun_out/r04p/seed_product.log 2>&1
python tests/diag_one_seed_next_ops.py 760460 profiles/variants/r03.so > gpurun_out/r04p/seed_r03.log 2>&1
unset DRTK_CAPI_POISON
python profiles/host_time.py --out gpurun_out/r04p/host_time.txt > gpurun_out/r04p/host_time.log 2>&1
tail -3 gpurun_out/r04p/seed_*.log; head -20 gpurun_out/r04p/host_time.txt
