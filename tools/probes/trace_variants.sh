# Diagnostic: kernel trace + stream-level timeline of one steady-state step for several engine-switch settings, one box.
# Usage on the GPU box: [BENCH_ARGS="--emulate-world 8"] bash tools/probes/trace_variants.sh <tag> "" "SDA_ENGINE_flat_tiles_backward=True" ...
TAG=$1; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
i=0
for S in "$@"; do
  cd /tmp && export TMPDIR=/tmp
  ( for kv in $S; do export $kv; done
    rocprofv3 --kernel-trace --output-format csv -d $O/trace$i -o tr -- python3 $R/bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg --no-feed-leg $BENCH_ARGS > $O/b$i.json 2> $O/err$i.txt )
  cd $R
  echo "== variant $i [$S]" > $O/timeline$i.txt
  python tools/timeline.py $O/trace$i/tr_kernel_trace.csv auto list >> $O/timeline$i.txt 2>&1
  head -7 $O/timeline$i.txt
  rm -rf $O/trace$i
  i=$((i+1))
done
