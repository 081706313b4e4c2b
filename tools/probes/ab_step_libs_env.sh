# Diagnostic: free-running step period for (library variant, engine-switch setting) pairs, alternating, on one box.
# Usage on the GPU box: bash tools/probes/ab_step_libs_env.sh rounds "new:" "v2:SDA_ENGINE_x=False" ...   (variant under variants/, then env settings)
cd $GRAFT_REPO_ROOT
R=$1; shift
for i in $(seq $R); do
  for P in "$@"; do
    V=${P%%:*}; S=${P#*:}
    cp variants/libsdamd_$V.so speech_decoding_amd/libsdamd.so
    echo "== $V [$S]: $(env $S timeout -k 10 120 python tools/step_series.py 60 4 2>/dev/null | tail -1)"
  done
done
