# Diagnostic: the same step / kernels with libsdamd.so variants built under variants/ (compiler scheduling strategies).
# Usage on the GPU box: bash tools/probes/variant_sweep.sh default max-ilp ...
set -e
cd $GRAFT_REPO_ROOT
for V in "$@"; do
  cp variants/libsdamd_$V.so speech_decoding_amd/libsdamd.so
  echo "== $V"
  timeout -k 10 120 python tools/bench_conv.py 2>/dev/null | grep -E " (full|flat|flat_noepi|plain) |wgrad " | grep -v perm
  timeout -k 10 120 python tools/step_series.py 60 4 2>/dev/null | tail -1
done
