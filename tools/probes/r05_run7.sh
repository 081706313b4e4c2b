cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05h
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "wgrad" > gpurun_out/r05h/tests.txt 2>&1; tail -5 gpurun_out/r05h/tests.txt
for V in "SDA_WGRAD_RING=1" "SDA_WGRAD_RING=0"; do echo "== $V"; env $V timeout -k 10 100 python tools/bench_wgrad.py 2>&1 | grep "k3"; done
bash tools/probes/ab_step_env.sh 2 "SDA_WGRAD_RING=1" "SDA_WGRAD_RING=0"
