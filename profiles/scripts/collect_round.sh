# Regenerates the per-round artefacts on the GPU box (everything lands in gpurun_out/round/):
#   pmc/{FETCH_SIZE,WRITE_SIZE}/  separate counter passes over profiles/kernel_bench.py
#   traffic.json                  HBM bytes per kernel launch from those passes (profiles/make_traffic.py)
#   bench_n1.json                 python bench.py                         (the driver's N=1 command)
#   stats/                        rocprofv3 --kernel-trace --stats of the same command (without the CPU-baseline leg and
#                                 without the graph-replay child process: nothing is spawned under the profiler)
# The counter passes come FIRST and their traffic.json is put where bench.py reads it (the newest profiles/rNN/ of
# this scratch copy of the repo), so the `traffic` in bench_n1.json is this collection's own figure and not the
# previously committed one.  Copy the summaries into profiles/rNN/ afterwards (see profiles/README.md).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/round
RDIR=$(ls -d profiles/r[0-9]* | sort | tail -1)
rm -rf $OUT; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc/$c -- python3 profiles/kernel_bench.py --reps 2 > $OUT/pmc_$c.log 2>&1
  python3 profiles/summarize_pmc.py $OUT/pmc/$c $OUT/kernel_bench_pmc_$c.txt > /dev/null
done
python3 profiles/make_traffic.py $OUT/pmc $OUT/traffic.json > /dev/null && cp $OUT/traffic.json $RDIR/traffic.json
python3 bench.py 2> $OUT/bench.err | grep "^{" > $OUT/bench_n1.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --cpu-sample-views 0 --no-graph > $OUT/stats.log 2>&1
python3 profiles/summarize_stats.py $OUT/stats $OUT/bench_step_kernel_stats.txt > /dev/null
cat $OUT/bench_n1.json
