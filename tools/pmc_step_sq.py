"""Per-kernel summary of the SQ counter passes of tools/pmc_step_sq.sh (diagnostic)."""
import collections, csv, glob, re, sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"]][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1


def short(k):
    k = k.replace("sda::(anonymous namespace)::", "").replace("sda::", "").replace("void ", "")
    k = k.replace("unsigned short", "bf16").replace("_Float16", "f16")
    return re.sub(r"\(.*$", "", k)[:64]


rows = []
for k, c in acc.items():
    if "sda" not in k:
        continue
    g = lambda n, c=c: c[n][0] if n in c else 0.0
    n = max(v[1] for v in c.values())
    rows.append((g("SQ_BUSY_CYCLES"), short(k), n, g, ))
rows.sort(reverse=True)
tot_busy = sum(r[0] for r in rows)
tot_mfma = sum(r[3]("SQ_VALU_MFMA_BUSY_CYCLES") for r in rows)
print("# SQ counters of the bench step (bf16, config 2), per kernel, summed over all launches of 5 steps; tools/pmc_step_sq.sh")
print("# util = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES);  per-MFMA ratios from SQ_INSTS_*;  wait = SQ_WAIT_ANY / SQ_WAVE_CYCLES;")
print("# conflicts = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE")
print(f"# STEP (all sda:: kernels): matrix-pipe utilisation {tot_mfma / (32 * tot_busy):.3f} of the cycles any kernel of the step keeps the chip busy")
print(f"{'kernel':64s} {'launches':>8s} {'busy share':>10s} {'MFMA util':>9s} {'SALU/MFMA':>9s} {'VALU/MFMA':>9s} {'LDS/MFMA':>8s} {'VMEM/MFMA':>9s} {'wait':>6s} {'conflicts':>9s}")
for busy, name, n, g in rows:
    mf = g("SQ_INSTS_MFMA")
    per = lambda x: f"{g(x) / mf:9.2f}" if mf else f"{'-':>9s}"
    util = g("SQ_VALU_MFMA_BUSY_CYCLES") / (32 * busy) if busy else 0.0
    wait = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else 0.0
    conf = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else 0.0
    print(f"{name:64s} {n:8d} {busy / tot_busy:10.3f} {util:9.3f} {per('SQ_INSTS_SALU')} {per('SQ_INSTS_VALU')} {per('SQ_INSTS_LDS')[1:]} {per('SQ_INSTS_VMEM')} {wait:6.2f} {conf:9.3f}")
