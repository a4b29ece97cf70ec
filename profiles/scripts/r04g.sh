mkdir -p gpurun_out/r04g; O=gpurun_out/r04g
for F in 0 8 1 32 9; do python profiles/shape_bench.py --what interp_c --channels 16 --grads both,attr_only --reps 20 --flags $F --out $O/interp_flags$F.json > /dev/null 2> $O/interp_flags$F.log; grep "'C'" $O/interp_flags$F.log | sed "s/^/flags $F: /" | cut -c1-90; done
python profiles/mipmap_bench.py --reps 10 --flags 0,15,31,47,16,32 > $O/mipmap_tiled_ablate.log 2>&1; grep flags $O/mipmap_tiled_ablate.log
