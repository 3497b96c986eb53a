# One gpurun call = one evidence set (same box): bench line, rocprofv3 kernel stats, HBM traffic (two PMC passes).
# (forward counts: bench.py runs 5 set-up forwards -- the third records the launch plan and replays it once for the self-check: 6 forwards
#  of kernels -- then W warm-up and K timed ones: --steps 5 --warmup 2 = 13 forwards, --steps 1 --warmup 1 = 8)
#   bash tools/collect_profiles.sh <tag>        -> gpurun_out/<tag>_{bench.json,kernel_stats.csv,pmc_hbm_traffic.json}
set -e
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench_err.txt
rm -rf gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-profile --no-host-io --no-configs > gpurun_out/${TAG}_stats.log 2>&1
python tools/pmc_summary.py stats gpurun_out/${TAG}_stats 13 gpurun_out/${TAG}_kernel_stats.csv > gpurun_out/${TAG}_stats_summary.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d gpurun_out/${TAG}_fetch --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host-io --no-configs > gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d gpurun_out/${TAG}_write --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile --no-host-io --no-configs > gpurun_out/${TAG}_write.log 2>&1
python tools/pmc_summary.py traffic gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write 8 gpurun_out/${TAG}_pmc_hbm_traffic.json > gpurun_out/${TAG}_pmc_summary.txt 2>&1
# keep only the summaries (raw traces are large)
rm -rf gpurun_out/${TAG}_stats gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write
