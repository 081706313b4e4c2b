# One round's measurement set, run on the GPU box:  bash tools/profile_round.sh r04
# (clean bench line, the same command under rocprofv3 --kernel-trace --stats, the two PMC passes behind roofline.traffic,
#  the fp32 line; the summaries worth keeping are copied from gpurun_out/<tag>/ into profiles/ by hand)
set -e
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
PART=${2:-all}      # 1 = the headline set, 2 = the round-5 additions (each fits one 20-minute gpurun call), all = both
if [ "$PART" != "2" ]; then
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/bench_bf16.json 2> $O/bench.err
tail -c 400 $O/bench_bf16.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --dtype fp32 --no-cpu-baseline > $O/bench_fp32.json 2> $O/bench_fp32.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o tr -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-sync-leg > $O/bench_under_rocprof.json 2> $O/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg > /dev/null 2> $O/pmc_write.err
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_summary.json | head -8
python tools/timeline.py $O/trace/tr_kernel_trace.csv auto list > $O/step_timeline.txt
head -6 $O/step_timeline.txt
# round 4 additions: SQ counters of the step per kernel (MFMA utilisation, SALU / VALU per MFMA), collate throughput
bash tools/pmc_step_sq.sh $TAG > /dev/null 2>&1 || true
head -12 $O/step_sq_counters.txt
timeout -k 10 120 python tools/bench_collate.py > $O/bench_collate.txt 2>&1 || true
cat $O/bench_collate.txt | grep -v amdgpu.ids
fi
if [ "$PART" != "1" ]; then
# round 5 additions: the per-rank step of an 8-rank job on the one GPU (collectives through RCCL at world size 1), the per-GPU
# shapes of configs[3] / configs[4], each as a driver-style JSON line plus the rocprofv3 kernel-stats CSV of the same command
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --emulate-world 8 --coll-timer-steps 3 > $O/bench_emul8.json 2> $O/bench_emul8.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --emulate-world 8 --emulate-no-copy --no-host-sync-leg > $O/bench_emul8_nocopy.json 2> $O/bench_emul8_nocopy.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --config 4 > $O/bench_config4.json 2> $O/bench_config4.err
timeout -k 10 300 python bench.py --steps 10 --warmup 4 --config 5 > $O/bench_config5.json 2> $O/bench_config5.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --dtype fp16 --no-cpu-baseline > $O/bench_fp16.json 2> $O/bench_fp16.err
cd /tmp
for V in "emul8:--emulate-world 8" "config4:--config 4" "config5:--config 5 --steps 6 --warmup 3"; do
  name=${V%%:*}; args=${V#*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$name -o tr -- python3 $R/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-host-sync-leg --no-feed-leg $args > $O/bench_${name}_under_rocprof.json 2> $O/trace_$name.err
  cp $O/trace_$name/tr_kernel_stats.csv $O/bench_${name}_kernel_stats.csv 2>/dev/null || true
  if [ "$name" = "emul8" ]; then python $R/tools/timeline.py $O/trace_$name/tr_kernel_trace.csv auto list > $O/step_timeline_emul8.txt 2>&1 || true; fi
  rm -rf $O/trace_$name
done
cd $R
tail -c 300 $O/bench_emul8.json; tail -c 300 $O/bench_config4.json; tail -c 300 $O/bench_config5.json

fi
