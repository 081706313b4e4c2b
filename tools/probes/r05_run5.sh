cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05f
for V in "SDA_ENGINE_flat_tiles_forward_fp32=True" "SDA_ENGINE_flat_tiles_forward_fp32=False"; do
echo "== $V"; env $V timeout -k 10 300 python bench.py --steps 10 --warmup 4 --dtype fp32 --no-cpu-baseline --no-host-sync-leg --no-feed-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fp32', d['ms_per_step'], d['roofline']['frac'], d['kernel_tflops'])"
done
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r05f/tests.txt 2>&1; tail -5 gpurun_out/r05f/tests.txt
