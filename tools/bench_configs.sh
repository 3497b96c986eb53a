cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-host-io --no-profile"
python bench.py --variant lite --height 256 --width 256 --steps 50 --warmup 10 $F > gpurun_out/cfg_c1.json 2>/dev/null
python bench.py --variant lite --height 256 --width 448 --global-off --steps 50 --warmup 10 $F > gpurun_out/cfg_c2.json 2>/dev/null
python bench.py --variant base --height 540 --width 960 --steps 20 --warmup 5 $F > gpurun_out/cfg_c3.json 2>/dev/null
python bench.py --variant base --height 1080 --width 1920 --steps 10 --warmup 3 $F > gpurun_out/cfg_c4.json 2>/dev/null
python bench.py --variant base --height 2160 --width 4096 --steps 4 --warmup 2 $F > gpurun_out/cfg_c5.json 2>/dev/null
python bench.py --variant base --height 1080 --width 1920 --precision f32 --steps 5 --warmup 2 $F > gpurun_out/cfg_c4f32.json 2>/dev/null
