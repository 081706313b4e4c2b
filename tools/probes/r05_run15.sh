cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py tests/test_train_gpu.py tests/test_dp_gpu.py -m gpu -q -x 2>&1 | tail -3
for V in "SDA_FEED_RECS=2" "SDA_FEED_RECS=7"; do
  echo "== $V"; env $V timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['with_feed']['ms_per_step'], d['with_feed']['host_enqueue_ms_per_step'])"
done
