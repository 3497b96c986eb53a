set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py > gpurun_out/v10_bench.json 2> gpurun_out/v10_bench_err.txt
rm -rf gpurun_out/v10_stats gpurun_out/v10_fetch gpurun_out/v10_write
rocprofv3 --kernel-trace --stats -d gpurun_out/v10_stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-host-io > gpurun_out/v10_stats.log 2>&1
python tools/pmc_summary.py stats gpurun_out/v10_stats 7 gpurun_out/v10_kernel_stats.csv > gpurun_out/v10_stats_summary.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/v10_fetch --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host-io > gpurun_out/v10_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/v10_write --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host-io > gpurun_out/v10_write.log 2>&1
python tools/pmc_summary.py traffic gpurun_out/v10_fetch gpurun_out/v10_write 2 gpurun_out/v10_pmc_traffic.json > gpurun_out/v10_pmc_summary.txt 2>&1
# keep only the summaries (raw traces are large)
rm -rf gpurun_out/v10_stats gpurun_out/v10_fetch gpurun_out/v10_write
