# Full evidence set of one build on one box (one gpurun call): bench line, rocprofv3 kernel stats, HBM traffic, SQ / LDS / cache
# counters, all configs.   bash tools/collect_all.sh <tag>
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh $TAG
rm -f gpurun_out/sq_summary.txt
bash tools/pmc_sq.sh > gpurun_out/pmc_sq.log 2>&1
bash tools/pmc_lds.sh > gpurun_out/pmc_lds.log 2>&1
bash tools/pmc_cache.sh > gpurun_out/pmc_cache.log 2>&1
bash tools/bench_configs.sh $TAG > gpurun_out/cfgs.log 2>&1
python tools/profile_layers.py > gpurun_out/${TAG}_layers.txt 2>&1
python tools/inflight_ab.py c1 c2 c3 > gpurun_out/${TAG}_inflight_ab.txt 2>&1
# every output that is committed must exist and be non-empty (an empty file was committed as evidence)
for f in ${TAG}_bench.json ${TAG}_kernel_stats.csv ${TAG}_pmc_hbm_traffic.json ${TAG}_layers.txt ${TAG}_bench_all_configs.json ${TAG}_inflight_ab.txt sq_summary.txt; do
  test -s gpurun_out/$f || { echo "collect_all: gpurun_out/$f is missing or empty"; exit 1; }
done
