# Round 6's committed evidence in one gpurun call (everything lands in gpurun_out/r06_final/; copy into profiles/r06/).
# Every file DESIGN.md / profiles/NOTES.md cite for round 6 comes from here (profiles/README.md maps file -> command):
#   collect_round.sh's artefacts   PMC passes, traffic.json (with the kernel sources' hashes), bench_n1.json, rocprofv3 kernel
#                                  stats of the same command
#   other_configs/*.json           bench.py --config 2..5 and the 1M-triangle geometry rows
#   interp_bwd_by_C.json           profiles/shape_bench.py: interpolate backward, C = 4 ... 64, f32 and f64 (the default route: padded
#                                  rows where attr_grad's rows are not whole 64-byte segments); interp_bwd_by_C_unpadded.json:
#                                  the same sweep of the float counts through drtk_amd_interpolate_backward (no workspace)
#   sparse_and_wireframe.txt       profiles/kernel_bench.py: interpolation_matrix (kernel alone / whole call) + backward, normal-matrix
#                                  values + backward, rasterize(wireframe=True) beside rasterize, at 100k and 250k triangles
#   mipmap_by_C.txt                the sampler's forward / backward on the textured inputs: C = 3 / 8 / 16, the three padding modes,
#                                  bicubic; mipmap_f64.txt; mipmap_minified.txt (kernel_bench's scenes)
#   textured_step_kernel_stats.txt rocprofv3 --kernel-trace --stats of the textured step
#   graph_vs_eager.txt             profiles/trace_gaps.py over kernel traces of the eager step and of the captured graph's replays
#   tile_raster_pmc_sq.txt, backward_kernels_pmc.txt   SQ counters of the FINAL kernels (separate --pmc passes)
#   host_time_config2.txt          profiles/host_time.py
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
F=gpurun_out/r06_final; rm -rf $F; mkdir -p $F
bash profiles/scripts/collect_round.sh > $F/collect_round.log 2>&1
cp gpurun_out/round/bench_n1.json gpurun_out/round/bench_step_kernel_stats.txt gpurun_out/round/traffic.json gpurun_out/round/kernel_bench_pmc_*.txt $F/ 2>/dev/null
bash profiles/scripts/other_configs.sh > $F/other_configs.log 2>&1
mkdir -p $F/other_configs; cp gpurun_out/configs/*.json $F/other_configs/
python3 profiles/shape_bench.py --what interp_c --reps 10 --channels 4,5,6,7,8,9,10,11,12,13,14,15,16,17,20,21,24,28,32,37,40,64 --dtypes f32,f64 --out $F/interp_bwd_by_C.json > /dev/null 2> $F/interp_bwd_by_C.log
DRTK_CAPI_NO_INTERP_WS=1 python3 profiles/shape_bench.py --what interp_c --reps 10 --channels 11,12,13,14,15,17,20,21,24,28,37,40 --dtypes f32 --out $F/interp_bwd_by_C_unpadded.json > /dev/null 2>> $F/interp_bwd_by_C.log
python3 profiles/shape_bench.py --what raster,f64 --reps 10 --split-dir $F > $F/shape_bench.json 2> $F/shape_bench.log
( python3 profiles/kernel_bench.py --only interpolation_matrix_kernel_only,interpolation_matrix,interpolation_matrix_backward,normal_matrix_values,normal_matrix_values_backward,rasterize_wireframe,rasterize --reps 5 2>&1 | grep -v amdgpu.ids
  echo "--mesh 250k"; python3 profiles/kernel_bench.py --only interpolation_matrix_kernel_only,normal_matrix_values,normal_matrix_values_backward,rasterize_wireframe,rasterize --mesh 250k --reps 5 2>&1 | grep -v amdgpu.ids ) > $F/sparse_and_wireframe.txt
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $F/tex_stats -- python3 bench.py --workload textured --no-graph --steps 10 --warmup 2 --cpu-sample-views 0 > $F/tex_stats.log 2>&1
python3 profiles/summarize_stats.py $F/tex_stats $F/textured_step_kernel_stats.txt > /dev/null 2>&1
rm -rf $F/tex_stats
( for c in 3 8 16; do python3 profiles/mipmap_bench.py --reps 5 --channels $c 2>&1 | grep "flags="; done
  for pad in zeros reflection; do python3 profiles/mipmap_bench.py --reps 5 --padding $pad 2>&1 | grep "flags="; done
  python3 profiles/mipmap_bench.py --reps 5 --channels 3 --bicubic 2>&1 | grep "flags=" ) > $F/mipmap_by_C.txt
( python3 profiles/mipmap_bench.py --reps 3 --f64 2>&1 | grep "flags="; python3 profiles/mipmap_bench.py --reps 3 --f64 --bicubic 2>&1 | grep "flags="; python3 profiles/mipmap_bench.py --reps 3 --f64 --channels 8 2>&1 | grep "flags=" ) > $F/mipmap_f64.txt
( for sc in 1.0 4.0; do echo "kernel_bench --uvscale $sc"; python3 profiles/kernel_bench.py --only mipmap_fwd,mipmap_bwd --reps 5 --uvscale $sc 2>&1 | grep -i "mipmap"; done ) > $F/mipmap_minified.txt
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $F/trace_eager -- python3 bench.py --no-graph --cpu-sample-views 0 --steps 8 --warmup 3 > $F/trace_eager.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $F/trace_graph -- python3 bench.py --graph-child --steps 8 > $F/trace_graph.log 2>&1
python3 profiles/trace_gaps.py $F/trace_eager $F/trace_graph --steps 5 > $F/graph_vs_eager.txt 2>&1
rm -rf $F/trace_eager $F/trace_graph
python3 profiles/host_time.py --out $F/host_time_config2.txt > /dev/null 2>&1
KERNELS=rasterize OUTDIR=pmc_raster bash profiles/scripts/pmc_backward.sh > /dev/null 2>&1; cp gpurun_out/pmc_raster/summary.txt $F/tile_raster_pmc_sq.txt
KERNELS=interpolate_backward,render_backward,edge_grad_backward_fused OUTDIR=pmc_bwd bash profiles/scripts/pmc_backward.sh > /dev/null 2>&1; cp gpurun_out/pmc_bwd/summary.txt $F/backward_kernels_pmc.txt
ls -la $F; cat $F/bench_n1.json | head -c 600; echo; cat $F/sparse_and_wireframe.txt $F/mipmap_by_C.txt $F/graph_vs_eager.txt
