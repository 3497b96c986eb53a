# Round-4 extras of the evidence set (one gpurun call): sweeps and same-process A/Bs quoted in DESIGN.md sections 3.7-3.9.
cd $GRAFT_REPO_ROOT
python tools/sweep_conv3p_wn.py c4 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_conv3p_wn_sweep.txt
python tools/sweep_conv3p_splitk.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_conv3p_splitk_sweep.txt
python tools/switch_ab.py use_splitk c1,c2,c3 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_splitk_ab.txt
python tools/switch_ab.py use_fused_tail c4,c3,c1 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_fused_tail_ab.txt
