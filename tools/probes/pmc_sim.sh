#!/bin/bash
# Diagnostic: SQ counter passes for sim_gemm alone.  usage: bash tools/probes/pmc_sim.sh <tag> Bm Bn
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/sq_$tag
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $out/p1 -o p1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/probes/sim_one.py "$@" > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL -d $out/p2 -o p2 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/probes/sim_one.py "$@" > /dev/null 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL -d $out/p3 -o p3 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/probes/sim_one.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    for f in glob.glob("$out/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "sim_gemm" not in r["Kernel_Name"]: continue
            a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for c, (v, n) in sorted(acc.items()):
            print("$tag %-32s %14.0f per launch (n=%d)" % (c, v / n, n))
PY
