cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -m gpu -q -x -k "param_gemm or e2e or composed or fixture" 2>&1 | tail -3
bash tools/probes/ab_step_env.sh 3 "" "SDA_PGEMM_K16=1"
