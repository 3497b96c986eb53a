# Experiment builds of the deferred-epilogue kernel (tools/lib/libatmvfi_hip_deexp<bits>.so, csrc/conv3p_de.inc ATMVFI_DE_EXP) against the
# product on the big layer shapes: where does a tile's time go?  bash tools/de_exp.sh "1 2 4 6"
cd $GRAFT_REPO_ROOT
for sh in "1 1088 1920 64 64 0" "1 1088 1920 101 101 0" "1 544 960 197 197 0"; do
  for e in "" $1; do
    lib=""; [ -n "$e" ] && lib=tools/lib/libatmvfi_hip_deexp$e.so
    echo "== exp ${e:-product}"
    ATMVFI_LIB=$lib ATMVFI_DE_NOCHECK=1 python tools/de_check.py $sh 2>&1 | grep -E "^N[0-9]"
  done
done
