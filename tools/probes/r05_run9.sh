cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05j
timeout -k 10 1100 python -m pytest tests -m gpu -q > gpurun_out/r05j/tests.txt 2>&1; tail -5 gpurun_out/r05j/tests.txt
