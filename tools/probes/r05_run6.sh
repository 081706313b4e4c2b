cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05g
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_loss_block_gpu.py -m gpu -q -x -k "clip_dz or loss_block" > gpurun_out/r05g/tests.txt 2>&1; tail -5 gpurun_out/r05g/tests.txt
timeout -k 10 100 python tools/bench_loss.py > gpurun_out/r05g/bench_loss.txt 2>&1; tail -8 gpurun_out/r05g/bench_loss.txt
for V in "SDA_FEED_DEBUG=" "SDA_FEED_DEBUG=discard"; do
echo "== $V"; env $V timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['with_feed']['ms_per_step'], d['with_feed']['host_enqueue_ms_per_step'])"
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --emulate-world 8 --no-host-sync-leg --no-kernel-timer 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('emul8', d['ms_per_step'])"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --emulate-world 8 --emulate-no-copy --no-host-sync-leg --no-kernel-timer 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('emul8 nocopy', d['ms_per_step'])"
