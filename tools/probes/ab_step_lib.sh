# Diagnostic: free-running step period (tools/step_series.py) for library variants under variants/, alternating, on one box.
# Usage on the GPU box: bash tools/probes/ab_step_lib.sh old new [rounds]
set -e
cd $GRAFT_REPO_ROOT
A=$1; B=$2; R=${3:-2}
for i in $(seq $R); do
  for V in $A $B; do
    cp variants/libsdamd_$V.so speech_decoding_amd/libsdamd.so
    echo "== $V: $(timeout -k 10 120 python tools/step_series.py 60 4 2>/dev/null | tail -1)"
  done
done
cp variants/libsdamd_$B.so speech_decoding_amd/libsdamd.so
