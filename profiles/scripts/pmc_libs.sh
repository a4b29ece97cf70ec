#!/bin/bash
# SQ counters of kernel $MATCH for each library in $VARIANTS (profiles/variants/<name>.so; `main` = the product library)
export TMPDIR=/tmp
out=gpurun_out/pmc_libs; rm -rf $out; mkdir -p $out
for lib in $VARIANTS; do
  if [ $lib = main ]; then L=""; else L="--lib profiles/variants/$lib.so"; fi
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM" "SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE SQ_INSTS_FLAT SQ_ACTIVE_INST_FLAT SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INSTS_SENDMSG"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --output-format csv -d $out/$lib/$i -- python3 profiles/kernel_bench.py --only ${ONLY:-rasterize} --reps 3 $L > $out/$lib.$i.log 2>&1
  done
  python3 profiles/summarize_pmc.py $out/$lib $out/$lib.txt ${MATCH:-tile_raster} > /dev/null; echo "== $lib"; cat $out/$lib.txt
done
