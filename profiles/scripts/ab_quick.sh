#!/bin/bash
# Times only: `main` and every library named in $VARIANTS (profiles/variants/<name>.so), two interleaved rounds, ONE box.
out=gpurun_out/ab_quick; mkdir -p $out; rm -f $out/times.log
for rep in 1 2; do
for lib in main $VARIANTS; do
  for mesh in ${MESHES:-100k 250k}; do
    if [ $lib = main ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
    echo -n "$lib $mesh " >> $out/times.log
    python profiles/kernel_bench.py --only ${ONLY:-rasterize} --reps 20 --mesh $mesh $L 2>&1 | grep "ms" | tr '\n' ' ' >> $out/times.log; echo >> $out/times.log
  done
done
done
cat $out/times.log
