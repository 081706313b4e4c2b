# Diagnostic: per-kernel times of the bench step under rocprofv3 for library variants: bash tools/probes/prof_lib.sh old new
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in "$@"; do
  cp $R/variants/libsdamd_$V.so $R/speech_decoding_amd/libsdamd.so
  rm -rf $R/gpurun_out/prof_$V
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$V -o tr -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-sync-leg --no-kernel-timer > $R/gpurun_out/prof_$V.json 2> $R/gpurun_out/prof_$V.err
  find $R/gpurun_out/prof_$V -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $R/gpurun_out/prof_${V}_kernel_stats.csv
  rm -rf $R/gpurun_out/prof_$V
done
