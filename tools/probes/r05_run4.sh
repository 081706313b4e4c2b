cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05e
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_train_gpu.py tests/test_loss_block_gpu.py tests/test_dp_gpu.py tests/test_rccl_gpu.py -m gpu -q -x -k "fill_zero or gather_samples or clip_merge or resident or loss_block or dp or rccl" > gpurun_out/r05e/tests.txt 2>&1; tail -5 gpurun_out/r05e/tests.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bf16', d['ms_per_step'], d['with_feed']['ms_per_step'])"
for V in "SDA_ENGINE_flat_tiles_forward_fp32=True" "SDA_ENGINE_flat_tiles_forward_fp32=False"; do
echo "== $V"; env $V timeout -k 10 300 python bench.py --steps 10 --warmup 4 --dtype fp32 --no-cpu-baseline --no-host-sync-leg --no-feed-leg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('fp32', d['ms_per_step'], d['roofline']['frac'], d['kernel_tflops'])"
done
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --emulate-world 8 --no-host-sync-leg --no-kernel-timer 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('emul8', d['ms_per_step'])"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --emulate-world 8 --emulate-no-copy --no-host-sync-leg --no-kernel-timer 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('emul8 nocopy', d['ms_per_step'])"
