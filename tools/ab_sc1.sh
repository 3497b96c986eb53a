cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for lib in "" tools/lib/libatmvfi_hip_plain.so; do
  echo "== lib ${lib:-product (sc1 stores in gemm_pp)}"
  ATMVFI_LIB=$lib ATMVFI_PROFILE_MIN_MS=9 python tools/profile_layers.py 2>&1 | grep -E "^total|linear_split|deconv2x2_split|conv2d_split" 
  rm -rf /tmp/f1
  ATMVFI_LIB=$lib rocprofv3 --pmc FETCH_SIZE -d /tmp/f1 --output-format csv -- python3 tools/profile_layers.py > /dev/null 2>&1
  PMC_TOP=3 python tools/pmc_lds.py /tmp/f1 | grep -A1 "gemm_pp_kernel\|conv3x3_planes_kernel"
done
