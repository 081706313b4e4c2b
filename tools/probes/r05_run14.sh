cd $GRAFT_REPO_ROOT
for i in 1 2; do
for V in "SDA_FEED_AT=backward" "SDA_FEED_AT=inline" "SDA_FEED_AT=start"; do
  echo "== $V"; env $V timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['with_feed']['ms_per_step'], d['with_feed']['host_enqueue_ms_per_step'])"
done; done
