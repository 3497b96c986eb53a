# Same-box A/B of two builds on the per-launch table of a 1080p forward:  bash tools/ab_lib.sh tools/lib/libatmvfi_hip_base.so [grep pattern]
cd $GRAFT_REPO_ROOT
PAT=${2:-"^total|x +[0-9]+ "}
for rep in 1 2; do
for lib in "" $1; do
  echo "== lib ${lib:-product}"
  ATMVFI_LIB=$lib ATMVFI_PROFILE_MIN_MS=9 python tools/profile_layers.py 2>&1 | grep -E "$PAT" | head -8
done
done
