for f in 0 2 4 8 12 16 32 62; do
  echo "== flags $f" >> gpurun_out/abl.txt
  ATMVFI_LIB=atm-vfi_amd/libatmvfi_hip_ablate.so ATMVFI_LEGACY_ORDER=$f ATMVFI_CONV3_SCHED=row timeout -k 10 120 python tools/profile_layers.py 2>&1 | grep "^conv3x3_f16x3 " >> gpurun_out/abl.txt || exit 1
done
