# Round 4's committed evidence in one gpurun call (everything lands in gpurun_out/r04_final/; copy into profiles/r04/):
#   collect_round.sh's artefacts (PMC passes, traffic.json, bench_n1.json, rocprofv3 kernel stats of the same command)
#   other_configs/*.json           bench.py --config 2..5 and the 1M-triangle geometry rows
#   interp_bwd_by_C.json, raster_regimes.json, f64_and_odd_width.json     profiles/shape_bench.py
#   textured_step_kernel_stats.txt rocprofv3 --kernel-trace --stats of the textured step
#   mipmap_by_C.txt                the sampler's forward / backward at C = 3, 8, 16 (+ bicubic at 3); mipmap_f64_now.txt: in double
#   host_time_config2.txt          profiles/host_time.py: where the eager step of BASELINE configs[1] spends its host time
#   mipmap_pmc_sq.txt, tile_raster_pmc_sq.txt    SQ counters (separate --pmc passes)
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
F=gpurun_out/r04_final; rm -rf $F; mkdir -p $F
bash profiles/scripts/collect_round.sh > $F/collect_round.log 2>&1
cp gpurun_out/round/bench_n1.json gpurun_out/round/bench_step_kernel_stats.txt gpurun_out/round/traffic.json gpurun_out/round/kernel_bench_pmc_*.txt $F/ 2>/dev/null
bash profiles/scripts/other_configs.sh > $F/other_configs.log 2>&1
mkdir -p $F/other_configs; cp gpurun_out/configs/*.json $F/other_configs/
python3 profiles/shape_bench.py --reps 10 --channels 4,8,12,16,20,24,32,40,64 --split-dir $F > $F/shape_bench.json 2> $F/shape_bench.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $F/tex_stats -- python3 bench.py --workload textured --no-graph --steps 10 --warmup 2 --cpu-sample-views 0 > $F/tex_stats.log 2>&1
python3 profiles/summarize_stats.py $F/tex_stats $F/textured_step_kernel_stats.txt > /dev/null 2>&1
rm -rf $F/tex_stats
( for c in 3 8 16; do python3 profiles/mipmap_bench.py --reps 5 --channels $c 2>&1 | grep "flags="; done; python3 profiles/mipmap_bench.py --reps 5 --channels 3 --bicubic 2>&1 | grep "flags=" ) > $F/mipmap_by_C.txt
( python3 profiles/mipmap_bench.py --reps 3 --f64 2>&1 | grep "flags="; python3 profiles/mipmap_bench.py --reps 3 --f64 --bicubic 2>&1 | grep "flags="; python3 profiles/mipmap_bench.py --reps 3 --f64 --channels 8 2>&1 | grep "flags=" ) > $F/mipmap_f64_now.txt
python3 profiles/host_time.py --out $F/host_time_config2.txt > /dev/null 2>&1
BENCH="profiles/mipmap_bench.py --reps 2" OUTDIR=pmc_mip bash profiles/scripts/pmc_backward.sh > /dev/null 2>&1; cp gpurun_out/pmc_mip/summary.txt $F/mipmap_pmc_sq.txt
KERNELS=rasterize OUTDIR=pmc_raster bash profiles/scripts/pmc_backward.sh > /dev/null 2>&1; cp gpurun_out/pmc_raster/summary.txt $F/tile_raster_pmc_sq.txt
KERNELS=interpolate_backward,edge_grad_backward_fused OUTDIR=pmc_bwd bash profiles/scripts/pmc_backward.sh > /dev/null 2>&1; cp gpurun_out/pmc_bwd/summary.txt $F/backward_kernels_pmc.txt
ls -la $F; cat $F/bench_n1.json | head -c 600; echo; cat $F/mipmap_by_C.txt
