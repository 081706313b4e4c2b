#!/bin/bash
# The scaling curve of the training step on ONE node, as the driver runs it, plus what each collective costs:
#     bash tools/scale_run.sh [out_dir] [steps] [warmup]
# emits one bench.py JSON line per N in {1, 2, 4, 8} (N <= the GPUs of the node) into <out_dir>/scale_N.json — the N > 1 lines
# carry `collectives_us_per_step` (HIP-event brackets around every torch.distributed call of 5 extra steps AFTER the timed
# region: SyncBN statistics all-reduces, the speech-row all-gather, the loss's row-statistics all-gather, gradient buckets) —
# and prints the weak-scaling table.  DESIGN.md §5 holds the predictions these numbers check: 0.4-0.8 ms per step of
# latency-bound BatchNorm all-reduces at N = 8, +1.1 ms of loss arithmetic per rank (global negatives), ~80 % efficiency.
# Needs as many GPUs as the largest N it runs; nothing here fakes a curve on fewer.
set -e
OUT=${1:-gpurun_out/scale}
STEPS=${2:-20}
WARM=${3:-5}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
NGPU=$(python3 -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  [ "$N" -le "$NGPU" ] || { echo "skipping N=$N: the node has $NGPU GPU(s)"; continue; }
  if [ "$N" -eq 1 ]; then
    python3 "$ROOT/bench.py" --gpus 1 --steps "$STEPS" --warmup "$WARM" --no-cpu-baseline > "$OUT/scale_1.json"
  else
    python3 -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      "$ROOT/bench.py" --gpus "$N" --steps "$STEPS" --warmup "$WARM" --coll-timer-steps 5 > "$OUT/scale_$N.json"
  fi
  tail -c 300 "$OUT/scale_$N.json"; echo
done
python3 - "$OUT" <<'PY'
import json, sys, glob, os
rows = {}
for f in sorted(glob.glob(os.path.join(sys.argv[1], "scale_*.json"))):
    d = json.loads(open(f).read().strip().split("\n")[-1])
    rows[d["n_gpus"]] = d
if 1 in rows:
    base = rows[1]["value"]
    print(f"{'N':>2} {'segments/s':>12} {'ms/step':>8} {'efficiency':>10}")
    for n, d in sorted(rows.items()):
        print(f"{n:>2} {d['value']:>12.0f} {d['ms_per_step']:>8.3f} {d['value'] / (n * base):>10.3f}")
        for k, v in d.get("collectives_us_per_step", {}).items():
            print(f"      {k:60s} {v['calls_per_step']:6.1f} calls  {v['us_per_step']:9.1f} us per step")
PY
