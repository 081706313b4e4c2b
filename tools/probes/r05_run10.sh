cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05k
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "conv1_wide" > gpurun_out/r05k/tests.txt 2>&1; tail -5 gpurun_out/r05k/tests.txt
bash tools/probes/ab_step_env.sh 2 "" "SDA_ENGINE_wide_1x1_forward=False SDA_ENGINE_wide_1x1_backward=False" "SDA_ENGINE_wide_1x1_backward=False" "SDA_ENGINE_wide_1x1_forward=False"
