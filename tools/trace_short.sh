cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02c; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o tr -- python3 $R/bench.py --steps 12 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg > $O/b.json 2> $O/err.txt
tail -c 300 $O/b.json
