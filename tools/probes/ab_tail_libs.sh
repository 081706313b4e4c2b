# Diagnostic: the tail products alone (tools/probes/bench_tail.py) and the free-running step for library variants under variants/
# Usage on the GPU box: bash tools/probes/ab_tail_libs.sh a b c
cd $GRAFT_REPO_ROOT
cp speech_decoding_amd/libsdamd.so /tmp/libsdamd_keep.so
for V in "$@"; do
  cp variants/libsdamd_$V.so speech_decoding_amd/libsdamd.so
  echo "== $V"; timeout -k 10 120 python tools/probes/bench_tail.py 2>/dev/null | grep -v amdgpu
done
bash tools/probes/ab_step_libs.sh 2 "$@"
cp /tmp/libsdamd_keep.so speech_decoding_amd/libsdamd.so
