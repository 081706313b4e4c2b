# Diagnostic: engine clock and package power while the step runs (is the matrix pipe's clock the nominal 2.4 GHz under load?)
# Usage on the GPU box: bash tools/probes/clocks_under_load.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/clocks; mkdir -p $O
sample() { for i in $(seq $1); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | tr '\n' ' '; echo; sleep 0.2; done; }
echo "== idle"; sample 3
timeout -k 10 200 python tools/step_series.py 1500 4 > $O/series.log 2>&1 &
P=$!
sleep 20
echo "== step running"; sample 12
wait $P
tail -1 $O/series.log
