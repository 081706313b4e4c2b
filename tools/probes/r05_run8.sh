cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05i
timeout -k 10 1100 python -m pytest tests -m gpu -q -x > gpurun_out/r05i/tests.txt 2>&1; tail -5 gpurun_out/r05i/tests.txt
bash tools/probes/ab_step_env.sh 2 "" 
