for f in 0 4 8 12 16 28; do
  echo "== flags $f" >> gpurun_out/abl_split.txt
  ATMVFI_LIB=atm-vfi_amd/libatmvfi_hip_ablate.so ATMVFI_SPLIT_DEBUG=$f timeout -k 10 120 python tools/profile_layers.py 2>&1 | grep "^linear_split \|^deconv2x2_split " >> gpurun_out/abl_split.txt || exit 1
done
