# Same-box A/B of dw-conv builds on the per-launch table of a 1080p forward:  bash tools/dw_ab.sh [variant libraries ...]
# (the sliding-window leg needs the diagnostic library: make -C atm-vfi_amd/csrc ablate  ->  tools/lib/libatmvfi_hip_ablate.so)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
for rep in 1 2; do
for lib in "" "$@"; do
  echo "== ${lib:-product}"
  ATMVFI_LIB=$lib ATMVFI_PROFILE_MIN_MS=9 python tools/profile_layers.py 2>&1 | grep -E "^total|dwconv3x3_gelu +x"
done
echo "== product, sliding-window kernel"
ATMVFI_LIB=tools/lib/libatmvfi_hip_ablate.so ATMVFI_DWCONV_ROWS=1 ATMVFI_PROFILE_MIN_MS=9 python tools/profile_layers.py 2>&1 | grep -E "^total|dwconv3x3_gelu +x"
done > gpurun_out/r05c/dw_ab.txt 2>&1
cat gpurun_out/r05c/dw_ab.txt
