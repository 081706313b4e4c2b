# Diagnostic: free-running step period for any number of library variants under variants/, alternating: bash tools/probes/ab_step_libs.sh rounds a b c ...
cd $GRAFT_REPO_ROOT
R=$1; shift
for i in $(seq $R); do
  for V in "$@"; do
    cp variants/libsdamd_$V.so speech_decoding_amd/libsdamd.so
    echo "== $V: $(timeout -k 10 120 python tools/step_series.py 60 4 2>/dev/null | tail -1)"
  done
done
