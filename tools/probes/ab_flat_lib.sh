# Diagnostic: conv3_flat timelines for library variants under variants/ (GPU box): bash tools/probes/ab_flat_lib.sh old new
set -e
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  cp variants/libsdamd_$V.so speech_decoding_amd/libsdamd.so
  echo "== $V  K loop only (flag 256), plain"
  timeout -k 10 60 python tools/flat_timeline.py 256 256 320 plain 2>/dev/null | grep -vE "^stamp|places"
  echo "== $V  full"
  timeout -k 10 60 python tools/flat_timeline.py 0 256 320 full 2>/dev/null | grep -vE "^stamp|places"
done
