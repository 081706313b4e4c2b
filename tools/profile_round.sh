set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02b
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
tail -c 600 $O/bench.json
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o tr -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-sync-leg > $O/bench_under_rocprof.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg > /dev/null 2> $O/pmc_write.err
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_summary.json | head -8
ls $O/trace/*/ | head
