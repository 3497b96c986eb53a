# rocprofv3 kernel stats of the small configurations (one gpurun call):  bash tools/small_stats.sh <tag> [configs...]
TAG=${1:-r04}; shift
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for c in ${@:-c1 c2 c3}; do
  rm -rf gpurun_out/${TAG}_${c}_stats
  rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_${c}_stats --output-format csv -- python3 bench.py --config $c --steps 30 --warmup 4 --no-cpu-baseline --no-profile --no-host-io --no-configs > gpurun_out/${TAG}_${c}_stats.log 2>&1
  python tools/pmc_summary.py stats gpurun_out/${TAG}_${c}_stats 40 gpurun_out/${TAG}_${c}_kernel_stats.csv > /dev/null 2>&1
  rm -rf gpurun_out/${TAG}_${c}_stats
done
