# Diagnostic: the emulated N-rank step (bench.py --emulate-world 8 --emulate-no-copy) for library variants under variants/, alternating on one box
# Usage on the GPU box: bash tools/probes/ab_emul_libs.sh rounds v1 v2 ...
cd $GRAFT_REPO_ROOT
R=$1; shift
for i in $(seq $R); do
  for V in "$@"; do
    cp variants/libsdamd_$V.so speech_decoding_amd/libsdamd.so
    echo "== $V: $(timeout -k 10 200 python bench.py --emulate-world 8 --emulate-no-copy --steps 20 --warmup 5 --no-host-sync-leg --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['host_enqueue_ms_per_step'])")"
  done
done
