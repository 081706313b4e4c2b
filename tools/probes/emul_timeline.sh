cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/emu2; mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 --no-host-sync-leg --no-cpu-baseline --no-feed-leg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=1', d['ms_per_step'])"
python bench.py --emulate-world 2 --emulate-no-copy --steps 20 --warmup 5 --no-host-sync-leg --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('emu2 nocopy', d['ms_per_step'])"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o tr -- python3 $R/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-host-sync-leg --no-feed-leg --emulate-world 2 --emulate-no-copy > $O/under.json 2> $O/trace.err
python $R/tools/timeline.py $O/trace/tr_kernel_trace.csv auto list > $O/timeline.txt 2>&1
head -40 $O/timeline.txt
