#!/bin/bash
# rocprofv3 kernel stats of kernel_bench --only $ONLY for each library in $VARIANTS
export TMPDIR=/tmp
out=gpurun_out/stats_libs; rm -rf $out; mkdir -p $out
for lib in $VARIANTS; do
  if [ $lib = main ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$lib -- python3 profiles/kernel_bench.py --only ${ONLY:-rasterize} --reps 20 $L > $out/$lib.log 2>&1
  python3 profiles/summarize_stats.py $out/$lib $out/$lib.txt > /dev/null; echo "== $lib"; grep -i "raster\|bin_\|fill_bytes" $out/$lib.txt | cut -c1-110
done
