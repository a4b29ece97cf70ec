# TA / TCP / TD / LDS utilisation of the backward kernels; one small counter set per pass, each
# pass bounded by its own timeout (a set the profiler rejects aborts instead of hanging the box).
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${OUTDIR:-pmc_ta}
mkdir -p $OUT
for set in "TA_TA_BUSY_sum TA_BUSY_avr GRBM_GUI_ACTIVE" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TD_TD_BUSY_sum TD_TC_STALL_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 150 rocprofv3 --pmc $set --output-format csv -d $OUT/$tag -- python3 ${BENCH:-profiles/kernel_bench.py --only ${KERNELS:-interpolate_backward} --reps 2} > $OUT/$tag.log 2>&1
  echo "$tag rc=$?"
done
python3 profiles/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
