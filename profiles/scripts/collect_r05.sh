# Round 5's committed evidence in one gpurun call (everything lands in gpurun_out/r05_final/; copy into profiles/r05/).
# Every file DESIGN.md / profiles/NOTES.md cite for round 5 comes from here (profiles/README.md maps file -> command):
#   collect_round.sh's artefacts   PMC passes, traffic.json (with the kernel sources' hashes), bench_n1.json, rocprofv3 kernel
#                                  stats of the same command
#   other_configs/*.json           bench.py --config 2..5 and the 1M-triangle geometry rows
#   interp_bwd_by_C.json           profiles/shape_bench.py: interpolate backward, C = 4 ... 64 incl. every count from 4 to 16, f32 and f64
#   raster_regimes.json, f64_and_odd_width.json                                      profiles/shape_bench.py
#   textured_step_kernel_stats.txt rocprofv3 --kernel-trace --stats of the textured step
#   mipmap_by_C.txt, mipmap_f64.txt, mipmap_minified.txt   the sampler's forward / backward (textured inputs; kernel_bench's scenes)
#   mipmap_ablation.txt, mipmap_rounds.txt   ablation build (python drtk_amd/build.py --ablation BEFORE the gpurun call)
#   mipmap_tile_times.txt          ablation build: per-tile timeline, phases of a tile, where the left-over pairs are
#                                  (mipmap_bench.py --tile-times / --tile-phases / --leftover-dump -> summarize_tile_times.py)
#   mipmap_atomic_requests.txt     TCP_TCC_ATOMIC_WITHOUT_RET_REQ of the lean backward under the ablation masks 0 / 256 / 8 / 12
#   mipmap_pmc_sq.txt, mipmap_pmc_ta.txt, tile_raster_pmc_sq.txt, backward_kernels_pmc.txt   SQ / TA counters (separate --pmc passes)
#   micro_valu_issue.txt           profiles/micro/valu_issue.hip
#   host_time_config2.txt          profiles/host_time.py
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
F=gpurun_out/r05_final; rm -rf $F; mkdir -p $F
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_issue profiles/micro/valu_issue.hip && timeout 300 /tmp/valu_issue > $F/micro_valu_issue.txt 2>&1
bash profiles/scripts/collect_round.sh > $F/collect_round.log 2>&1
cp gpurun_out/round/bench_n1.json gpurun_out/round/bench_step_kernel_stats.txt gpurun_out/round/traffic.json gpurun_out/round/kernel_bench_pmc_*.txt $F/ 2>/dev/null
bash profiles/scripts/other_configs.sh > $F/other_configs.log 2>&1
mkdir -p $F/other_configs; cp gpurun_out/configs/*.json $F/other_configs/
python3 profiles/shape_bench.py --what interp_c --reps 10 --channels 4,5,6,7,8,9,10,11,12,13,14,15,16,17,20,21,24,32,37,40,64 --dtypes f32,f64 --out $F/interp_bwd_by_C.json > /dev/null 2> $F/interp_bwd_by_C.log
python3 profiles/shape_bench.py --what raster,f64 --reps 10 --split-dir $F > $F/shape_bench.json 2> $F/shape_bench.log
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $F/tex_stats -- python3 bench.py --workload textured --no-graph --steps 10 --warmup 2 --cpu-sample-views 0 > $F/tex_stats.log 2>&1
python3 profiles/summarize_stats.py $F/tex_stats $F/textured_step_kernel_stats.txt > /dev/null 2>&1
rm -rf $F/tex_stats
( for c in 3 8 16; do python3 profiles/mipmap_bench.py --reps 5 --channels $c 2>&1 | grep "flags="; done; python3 profiles/mipmap_bench.py --reps 5 --channels 3 --bicubic 2>&1 | grep "flags=" ) > $F/mipmap_by_C.txt
( python3 profiles/mipmap_bench.py --reps 3 --f64 2>&1 | grep "flags="; python3 profiles/mipmap_bench.py --reps 3 --f64 --bicubic 2>&1 | grep "flags="; python3 profiles/mipmap_bench.py --reps 3 --f64 --channels 8 2>&1 | grep "flags=" ) > $F/mipmap_f64.txt
( for sc in 1.0 4.0; do echo "kernel_bench --uvscale $sc"; python3 profiles/kernel_bench.py --only mipmap_fwd,mipmap_bwd --reps 5 --uvscale $sc 2>&1 | grep -i "mipmap"; done ) > $F/mipmap_minified.txt
python3 profiles/mipmap_bench.py --reps 5 --flags 0,1,2,4,8,16,32,256,15,63 2>&1 | grep "flags=" > $F/mipmap_ablation.txt
python3 profiles/mipmap_bench.py --reps 2 --stats --rounds-stats 2>&1 | grep -i "taps/pixel\|lod:\|distinct\|tiles with" > $F/mipmap_rounds.txt
python3 profiles/host_time.py --out $F/host_time_config2.txt > /dev/null 2>&1
python3 profiles/mipmap_bench.py --reps 2 --tile-times /tmp/tt.npz > /dev/null 2>&1; python3 profiles/mipmap_bench.py --reps 2 --tile-phases --tile-times /tmp/tp.npz > /dev/null 2>&1
python3 profiles/mipmap_bench.py --reps 2 --leftover-dump /tmp/tl.npz > /dev/null 2>&1; python3 profiles/summarize_tile_times.py /tmp/tt.npz /tmp/tp.npz /tmp/tl.npz > $F/mipmap_tile_times.txt 2>&1
( for f in 0 256 8 12; do rm -rf gpurun_out/pmc_req; timeout 200 rocprofv3 --pmc TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum --output-format csv -d gpurun_out/pmc_req -- python3 profiles/mipmap_bench.py --flags $f --reps 2 > /dev/null 2>&1
  echo "flags=$f"; python3 profiles/summarize_pmc.py gpurun_out/pmc_req 2>&1 | grep -A1 "mipmap_backward_lean" | cut -c1-160; done ) > $F/mipmap_atomic_requests.txt
BENCH="profiles/mipmap_bench.py --reps 2" OUTDIR=pmc_mip bash profiles/scripts/pmc_backward.sh > /dev/null 2>&1; python3 profiles/summarize_pmc.py gpurun_out/pmc_mip $F/mipmap_pmc_sq.txt mipmap > /dev/null 2>&1
BENCH="profiles/mipmap_bench.py --reps 2" OUTDIR=pmc_mip_ta bash profiles/scripts/pmc_ta.sh > /dev/null 2>&1; python3 profiles/summarize_pmc.py gpurun_out/pmc_mip_ta $F/mipmap_pmc_ta.txt mipmap > /dev/null 2>&1
KERNELS=rasterize OUTDIR=pmc_raster bash profiles/scripts/pmc_backward.sh > /dev/null 2>&1; cp gpurun_out/pmc_raster/summary.txt $F/tile_raster_pmc_sq.txt
KERNELS=interpolate_backward,edge_grad_backward_fused OUTDIR=pmc_bwd bash profiles/scripts/pmc_backward.sh > /dev/null 2>&1; cp gpurun_out/pmc_bwd/summary.txt $F/backward_kernels_pmc.txt
ls -la $F; cat $F/bench_n1.json | head -c 600; echo; cat $F/mipmap_by_C.txt
