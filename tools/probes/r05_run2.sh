cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05c
timeout -k 10 600 python -m pytest tests/test_collate.py -m gpu -x -q > gpurun_out/r05c/collate.txt 2>&1; tail -3 gpurun_out/r05c/collate.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer > gpurun_out/r05c/bench_feed.json 2> gpurun_out/r05c/bench_feed.err; tail -c 900 gpurun_out/r05c/bench_feed.json
BENCH_ARGS="--emulate-world 8" bash tools/probes/trace_variants.sh r05c ""
