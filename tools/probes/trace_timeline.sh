# Diagnostic: kernel trace of the bench command and the stream-level view of one steady-state step (with the full listing).
# Usage on the GPU box: bash tools/probes/trace_timeline.sh <tag>
TAG=${1:-tl}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o tr -- python3 $R/bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg > $O/b.json 2> $O/err.txt
cd $R
python tools/timeline.py $O/trace/tr_kernel_trace.csv -6 list > $O/timeline.txt
head -8 $O/timeline.txt
rm -rf $O/trace
