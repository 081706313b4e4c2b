#!/bin/bash
# SQ counters of the BENCH STEP, per kernel (rocprofv3 --pmc, two passes of `bench.py --steps 3 --warmup 2`; GPU box):
#     bash tools/pmc_step_sq.sh r04      ->  gpurun_out/r04/step_sq_counters.txt  (copy into profiles/ by hand)
# MFMA utilisation of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES): the first counts matrix-pipe cycles summed
# over the chip's 1024 SIMDs, the second busy cycles summed over its 32 shader engines (MI355X_MICROARCH.md, cycle constants).
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
ARGS="--steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS \
  --output-format csv -d $O/sq1 -o p -- python3 $R/bench.py $ARGS > /dev/null 2> $O/sq1.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA \
  --output-format csv -d $O/sq2 -o p -- python3 $R/bench.py $ARGS > /dev/null 2> $O/sq2.err
python3 $R/tools/pmc_step_sq.py $O/sq1 $O/sq2 > $O/step_sq_counters.txt
rm -rf $O/sq1 $O/sq2
head -40 $O/step_sq_counters.txt
