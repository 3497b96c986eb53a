# SQ counters of the window-attention kernel alone (5 launches): bash tools/pmc_attn.sh local|global
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
W=${1:-local}
i=0
rm -f gpurun_out/attn_sq_$W.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM" "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_MISC" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum"; do
  i=$((i+1)); rm -rf gpurun_out/asq$i
  rocprofv3 --pmc $set -d gpurun_out/asq$i --output-format csv -- python3 tools/run_attn_once.py $W > gpurun_out/asq$i.log 2>&1 || echo "pass $i failed" >> gpurun_out/attn_sq_$W.txt
done
python tools/pmc_lds.py gpurun_out/asq1 gpurun_out/asq2 gpurun_out/asq3 gpurun_out/asq4 gpurun_out/asq5 gpurun_out/asq6 gpurun_out/asq7 gpurun_out/asq8 >> gpurun_out/attn_sq_$W.txt 2>&1
rm -rf gpurun_out/asq1 gpurun_out/asq2 gpurun_out/asq3 gpurun_out/asq4 gpurun_out/asq5 gpurun_out/asq6 gpurun_out/asq7 gpurun_out/asq8
