cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05m
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_loss_block_gpu.py -m gpu -q -x -k "clip or loss_block" > gpurun_out/r05m/tests.txt 2>&1; tail -3 gpurun_out/r05m/tests.txt
timeout -k 10 100 python tools/bench_loss.py 2>&1 | grep -E "Bm=|dZ gemm"
bash tools/probes/ab_step_env.sh 2 "SDA_DZ_TILES_MIN=257" "SDA_X=0"
