cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
rm -rf gpurun_out/lds1 gpurun_out/lds2 gpurun_out/lds3
F="--steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host-io --no-configs"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d gpurun_out/lds1 --output-format csv -- python3 bench.py $F > gpurun_out/lds1.log 2>&1
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d gpurun_out/lds2 --output-format csv -- python3 bench.py $F > gpurun_out/lds2.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d gpurun_out/lds3 --output-format csv -- python3 bench.py $F > gpurun_out/lds3.log 2>&1
python tools/pmc_lds.py gpurun_out/lds1 gpurun_out/lds2 gpurun_out/lds3 > gpurun_out/lds_summary.txt 2>&1
rm -rf gpurun_out/lds1 gpurun_out/lds2 gpurun_out/lds3
