# BENCH="profiles/mipmap_bench.py --reps 2" OUTDIR=pmc_mip bash profiles/scripts/pmc_backward.sh: the same passes over another driver
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/${OUTDIR:-pmc_ibwd}
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 200 rocprofv3 --pmc $set --output-format csv -d gpurun_out/${OUTDIR:-pmc_ibwd}/$tag -- python3 ${BENCH:-profiles/kernel_bench.py --only ${KERNELS:-interpolate_backward,render_backward,edge_grad_backward_fused} --reps 2} > gpurun_out/${OUTDIR:-pmc_ibwd}/$tag.log 2>&1
done
python3 profiles/summarize_pmc.py gpurun_out/${OUTDIR:-pmc_ibwd} > gpurun_out/${OUTDIR:-pmc_ibwd}/summary.txt 2>&1
tail -5 gpurun_out/${OUTDIR:-pmc_ibwd}/*.log
