set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py > gpurun_out/v12_bench.json 2> gpurun_out/v12_bench_err.txt
rm -rf gpurun_out/v12_stats gpurun_out/v12_fetch gpurun_out/v12_write
rocprofv3 --kernel-trace --stats -d gpurun_out/v12_stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-host-io > gpurun_out/v12_stats.log 2>&1
python tools/pmc_summary.py stats gpurun_out/v12_stats 7 gpurun_out/v12_kernel_stats.csv > gpurun_out/v12_stats_summary.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/v12_fetch --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host-io > gpurun_out/v12_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/v12_write --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host-io > gpurun_out/v12_write.log 2>&1
python tools/pmc_summary.py traffic gpurun_out/v12_fetch gpurun_out/v12_write 2 gpurun_out/v12_pmc_traffic.json > gpurun_out/v12_pmc_summary.txt 2>&1
# keep only the summaries (raw traces are large)
rm -rf gpurun_out/v12_stats gpurun_out/v12_fetch gpurun_out/v12_write
