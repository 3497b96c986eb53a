# Every BASELINE configuration as its own bench.py run + the exact-fp32 engine at c4, assembled into ONE json:
#   bash tools/bench_configs.sh <tag>   -> gpurun_out/<tag>_bench_all_configs.json   (fails if any run left no line)
TAG=${1:-r06}
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-host-io --no-profile --no-configs"
python bench.py --config c1 --steps 200 --warmup 20 $F > gpurun_out/cfg_c1.json 2>/dev/null
python bench.py --config c2 --steps 200 --warmup 20 $F > gpurun_out/cfg_c2.json 2>/dev/null
python bench.py --config c3 --steps 60 --warmup 6 $F > gpurun_out/cfg_c3.json 2>/dev/null
python bench.py --variant base --height 1080 --width 1920 --steps 20 --warmup 3 $F > gpurun_out/cfg_c4.json 2>/dev/null
python bench.py --config c5 --steps 4 --warmup 2 $F > gpurun_out/cfg_c5.json 2>/dev/null
python bench.py --variant base --height 1080 --width 1920 --precision f32 --steps 5 --warmup 2 $F > gpurun_out/cfg_c4f32.json 2>/dev/null
python - "$TAG" <<'PY'
import json, sys
out = {}
for c in ("c1", "c2", "c3", "c4", "c5", "c4f32"):
    txt = open(f"gpurun_out/cfg_{c}.json").read().strip()
    if not txt:
        raise SystemExit(f"bench_configs: the {c} run printed no line")
    out[c] = json.loads(txt.splitlines()[-1])
    out[c].pop("per_step_gpu_ms", None)
path = f"gpurun_out/{sys.argv[1]}_bench_all_configs.json"
json.dump(out, open(path, "w"), indent=1)
print(path, {c: v["value"] for c, v in out.items()})
PY
