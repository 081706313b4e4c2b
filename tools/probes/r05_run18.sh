cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05n
timeout -k 10 1150 python -m pytest tests -m gpu -q > gpurun_out/r05n/tests.txt 2>&1; tail -5 gpurun_out/r05n/tests.txt
