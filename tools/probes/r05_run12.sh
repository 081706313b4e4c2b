cd $GRAFT_REPO_ROOT
for V in "SDA_X=0" "SDA_DZ_TILES_MIN=256" "SDA_X=0" "SDA_DZ_TILES_MIN=256"; do echo "== $V"; env $V timeout -k 10 100 python tools/bench_loss.py 2>&1 | grep -E "Bm=|dZ gemm"; done
