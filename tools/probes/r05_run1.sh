set -x
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r05b
timeout -k 10 400 python -m pytest tests/test_collate.py -m gpu -x -q > gpurun_out/r05b/collate.txt 2>&1; tail -5 gpurun_out/r05b/collate.txt
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05b/bench_feed.json 2> gpurun_out/r05b/bench_feed.err; tail -c 1500 gpurun_out/r05b/bench_feed.json; tail -3 gpurun_out/r05b/bench_feed.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --emulate-world 8 --coll-timer-steps 3 > gpurun_out/r05b/bench_emul8.json 2> gpurun_out/r05b/bench_emul8.err; tail -c 2500 gpurun_out/r05b/bench_emul8.json; tail -3 gpurun_out/r05b/bench_emul8.err
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --config 4 > gpurun_out/r05b/bench_c4.json 2> gpurun_out/r05b/bench_c4.err; tail -c 1200 gpurun_out/r05b/bench_c4.json; tail -3 gpurun_out/r05b/bench_c4.err
timeout -k 10 300 python bench.py --steps 10 --warmup 4 --config 5 > gpurun_out/r05b/bench_c5.json 2> gpurun_out/r05b/bench_c5.err; tail -c 1200 gpurun_out/r05b/bench_c5.json; tail -3 gpurun_out/r05b/bench_c5.err
