"""Print one training step's kernel timeline from a rocprofv3 --kernel-trace CSV (diagnostic)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find step boundaries: the pack_rows kernel of X starts a step (first kernel of forward)
idx = [i for i, r in enumerate(rows) if "sa_fwd_partial" in r["Kernel_Name"]]
step = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) - 3
lo, hi = idx[step], idx[step + 1]
t0 = int(rows[lo]["Start_Timestamp"])
prev_end = {}
tot = collections.defaultdict(float)
for r in rows[lo:hi]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    q = r.get("Queue_Id", "?")
    name = r["Kernel_Name"].replace("void sda::", "").replace("sda::", "")[:60]
    gap = s - prev_end.get(q, s)
    prev_end[q] = e
    tot[name] += (e - s) / 1e3
    print(f"{s/1e3:9.1f} {(e-s)/1e3:8.1f} gap {gap/1e3:7.1f} q{q} {name}")
print("step span us:", (int(rows[hi]["Start_Timestamp"]) - t0) / 1e3)
