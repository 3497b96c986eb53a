cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
F="--steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host-io --no-configs"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VMEM" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC"; do
  i=$((i+1)); rm -rf gpurun_out/sq$i
  rocprofv3 --pmc $set -d gpurun_out/sq$i --output-format csv -- python3 bench.py $F > gpurun_out/sq$i.log 2>&1 || echo "pass $i failed" >> gpurun_out/sq_summary.txt
done
python tools/pmc_lds.py gpurun_out/sq1 gpurun_out/sq2 gpurun_out/sq3 gpurun_out/sq4 gpurun_out/sq5 >> gpurun_out/sq_summary.txt 2>&1
rm -rf gpurun_out/sq1 gpurun_out/sq2 gpurun_out/sq3 gpurun_out/sq4 gpurun_out/sq5
