# Regenerates the per-round artefacts on the GPU box (everything lands in gpurun_out/round/):
#   bench_n1.json                 python bench.py                         (the driver's N=1 command)
#   stats/                        rocprofv3 --kernel-trace --stats of the same command
#   pmc/{FETCH_SIZE,WRITE_SIZE}/  separate counter passes over profiles/kernel_bench.py
# Copy the summaries into profiles/rNN/ afterwards (see profiles/README.md).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/round
rm -rf $OUT; mkdir -p $OUT
python3 bench.py 2> $OUT/bench.err | grep "^{" > $OUT/bench_n1.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --cpu-sample-views 0 > $OUT/stats.log 2>&1
python3 profiles/summarize_stats.py $OUT/stats $OUT/bench_step_kernel_stats.txt > /dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc/$c -- python3 profiles/kernel_bench.py --reps 2 > $OUT/pmc_$c.log 2>&1
  python3 profiles/summarize_pmc.py $OUT/pmc/$c $OUT/kernel_bench_pmc_$c.txt > /dev/null
done
python3 profiles/make_traffic.py $OUT/pmc $OUT/traffic.json > /dev/null
cat $OUT/bench_n1.json
