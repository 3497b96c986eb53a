cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-host-io --no-profile --no-configs"
python bench.py --config c1 --steps 50 --warmup 10 $F > gpurun_out/cfg_c1.json 2>/dev/null
python bench.py --config c2 --steps 50 --warmup 10 $F > gpurun_out/cfg_c2.json 2>/dev/null
python bench.py --config c3 --steps 20 --warmup 5 $F > gpurun_out/cfg_c3.json 2>/dev/null
python bench.py --variant base --height 1080 --width 1920 --steps 10 --warmup 3 $F > gpurun_out/cfg_c4.json 2>/dev/null
python bench.py --config c5 --steps 4 --warmup 2 $F > gpurun_out/cfg_c5.json 2>/dev/null
python bench.py --variant base --height 1080 --width 1920 --precision f32 --steps 5 --warmup 2 $F > gpurun_out/cfg_c4f32.json 2>/dev/null
