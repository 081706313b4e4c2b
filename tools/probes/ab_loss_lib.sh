# Diagnostic: loss-stage micro-benchmarks (tools/bench_loss.py) for library variants under variants/: bash tools/probes/ab_loss_lib.sh a b
cd $GRAFT_REPO_ROOT
for V in "$@" "$@"; do
  cp variants/libsdamd_$V.so speech_decoding_amd/libsdamd.so
  echo "== $V"; timeout -k 10 100 python tools/bench_loss.py 2>/dev/null | grep -E "Bm=|sim gemm|dZ gemm"
done
