# Full evidence set of one build on one box (one gpurun call): bench line, rocprofv3 kernel stats, HBM traffic, SQ / LDS / cache
# counters, all configs.   bash tools/collect_all.sh <tag>
TAG=${1:-r04}
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh $TAG
rm -f gpurun_out/sq_summary.txt
bash tools/pmc_sq.sh > gpurun_out/pmc_sq.log 2>&1
bash tools/pmc_lds.sh > gpurun_out/pmc_lds.log 2>&1
bash tools/pmc_cache.sh > gpurun_out/pmc_cache.log 2>&1
bash tools/bench_configs.sh > gpurun_out/cfgs.log 2>&1
python tools/profile_layers.py > gpurun_out/${TAG}_layers.txt 2>&1
