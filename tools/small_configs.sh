# Small-frame configurations (BASELINE C1-C3) on one box: bench lines with launch plans on and off, from a HIP graph, and the
# rocprofv3 kernel stats of C1 / C2 (what the GPU itself spends per launch when the host is out of the way).
#   bash tools/small_configs.sh <tag>
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
F="--no-cpu-baseline --no-host-io --no-profile"
for c in c1 c2 c3; do
  python bench.py --config $c --steps 100 --warmup 20 $F > gpurun_out/${TAG}_${c}_plan.json 2>/dev/null
  python bench.py --config $c --steps 100 --warmup 20 --no-plans $F > gpurun_out/${TAG}_${c}_direct.json 2>/dev/null
  python bench.py --config $c --steps 100 --warmup 20 --graph $F > gpurun_out/${TAG}_${c}_graph.json 2>/dev/null
done
for c in c1 c2; do
  rm -rf gpurun_out/${TAG}_st_$c
  rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_st_$c --output-format csv -- python3 bench.py --config $c --steps 20 --warmup 5 $F > gpurun_out/${TAG}_st_$c.log 2>&1
  python tools/pmc_summary.py stats gpurun_out/${TAG}_st_$c 25 gpurun_out/${TAG}_${c}_kernel_stats.csv > gpurun_out/${TAG}_${c}_stats_summary.txt 2>&1
  rm -rf gpurun_out/${TAG}_st_$c
done
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_c?_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split("/")[-1], d["value"], "fps", d["ms_per_step"], "ms")
    except Exception as e: print(f, "FAILED", e)
PY
