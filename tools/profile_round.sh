# One round's measurement set, run on the GPU box:  bash tools/profile_round.sh r04
# (clean bench line, the same command under rocprofv3 --kernel-trace --stats, the two PMC passes behind roofline.traffic,
#  the fp32 line; the summaries worth keeping are copied from gpurun_out/<tag>/ into profiles/ by hand)
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_bf16.json 2> $O/bench.err
tail -c 400 $O/bench_bf16.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --dtype fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o tr -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-sync-leg > $O/bench_under_rocprof.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg > /dev/null 2> $O/pmc_write.err
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_summary.json | head -8
python tools/timeline.py $O/trace/tr_kernel_trace.csv > $O/step_timeline.txt
head -6 $O/step_timeline.txt
# round 4 additions: SQ counters of the step per kernel (MFMA utilisation, SALU / VALU per MFMA), collate throughput
bash tools/pmc_step_sq.sh $TAG > /dev/null 2>&1 || true
head -12 $O/step_sq_counters.txt
timeout -k 10 120 python tools/bench_collate.py > $O/bench_collate.txt 2>&1 || true
cat $O/bench_collate.txt | grep -v amdgpu.ids
