# Diagnostic: free-running step period for engine-switch settings (SDA_ENGINE_<switch>=<literal>), alternating, on one box.
# Usage on the GPU box: bash tools/probes/ab_step_env.sh rounds "A=1 B=2" "A=0"   (each argument = one setting; "" = defaults)
cd $GRAFT_REPO_ROOT
R=$1; shift
for i in $(seq $R); do
  for S in "$@"; do
    echo "== [$S]: $(env $S timeout -k 10 120 python tools/step_series.py 60 4 2>/dev/null | tail -1)"
  done
done
