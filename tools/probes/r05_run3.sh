cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05d
timeout -k 10 900 python -m pytest tests/test_collate.py tests/test_train_gpu.py -m gpu -q -k "collate or resident" > gpurun_out/r05d/collate.txt 2>&1; tail -3 gpurun_out/r05d/collate.txt
for V in "SDA_FEED_AT=backward SDA_FEED_PRIO=1" "SDA_FEED_AT=start SDA_FEED_PRIO=1" "SDA_FEED_AT=backward SDA_FEED_PRIO=0" "SDA_FEED_AT=start SDA_FEED_PRIO=0"; do
  echo "== $V"; env $V timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['with_feed']['ms_per_step'])"
done
