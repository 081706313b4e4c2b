cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_e2e_gpu.py tests/test_stream_race_gpu.py tests/test_fullsize_gpu.py -m gpu -q -x 2>&1 | tail -3
bash tools/probes/ab_step_env.sh 3 "" "SDA_ENGINE_bias_sums_at_end=False"
