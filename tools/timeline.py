"""Stream-level view of one training step from a rocprofv3 --kernel-trace CSV (diagnostic):
busy time per HIP stream, time with no / one / several kernels in flight, the largest gaps of the main stream, and per
kernel family the time it spends alone vs beside a kernel of another stream.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python3 bench.py --steps 6 --warmup 4 --no-cpu-baseline --no-kernel-timer --no-host-sync-leg
    python tools/timeline.py gpurun_out/tr/*/*_kernel_trace.csv"""
import collections, csv, re, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if "adam_multi" in r["Kernel_Name"]]
# which step: the given index, or by default the SHORTEST of the timed region's steps (the last ones are instrumented, and under
# the profiler the host stalls for milliseconds every few steps — a step with such a hole says nothing about the schedule)
if len(sys.argv) > 2 and sys.argv[2] not in ("", "auto"):
    k = int(sys.argv[2])
else:
    cand = range(max(1, len(ad) - 16), len(ad) - 4)
    k = min(cand, key=lambda i: int(rows[ad[i]]["End_Timestamp"]) - int(rows[ad[i - 1] + 1]["Start_Timestamp"])) - len(ad)
lo, hi = ad[k - 1] + 1, ad[k] + 1
t0, t1 = int(rows[lo]["Start_Timestamp"]), int(rows[hi - 1]["End_Timestamp"])
print(f"step span {(t1 - t0) / 1e3:.1f} us, {hi - lo} kernels")
def short(n):
    n = n.replace("void sda::", "").replace("sda::", "")
    m = re.match(r"([a-z_0-9A-Z]+)<([^>]*)>", n)
    return (m.group(1) + "<" + m.group(2).replace("unsigned short", "bf16").replace(" ", "") + ">") if m else n.split("(")[0][:40]
ks = [(int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0, r["Stream_Id"], short(r["Kernel_Name"])) for r in rows[lo:hi]]
bys = collections.defaultdict(list)
for s, e, q, n in ks: bys[q].append((s, e, n))
for q, v in sorted(bys.items(), key=lambda kv: -len(kv[1])):
    print(f"stream {q}: {len(v)} kernels, busy {sum(e - s for s, e, _ in v) / 1e3:.0f} us, active {v[0][0] / 1e3:.0f}..{v[-1][1] / 1e3:.0f}")
ev = sorted([(s, 1) for s, e, q, n in ks] + [(e, -1) for s, e, q, n in ks])
cur = last = 0; hist = collections.Counter()
for t, d in ev:
    hist[min(cur, 2)] += t - last; cur += d; last = t
print("no kernel in flight %.0f us, exactly one %.0f us, two or more %.0f us" % tuple(hist[i] / 1e3 for i in range(3)))
main = max(bys.values(), key=len)
gaps = sorted([((main[i + 1][0] - main[i][1]) / 1e3, main[i][2], main[i + 1][2]) for i in range(len(main) - 1)], reverse=True)
print("main stream: gaps sum %.0f us; largest:" % sum(g for g, _, _ in gaps if g > 0))
for g, a, b in gaps[:8]: print(f"   {g:7.1f} us  {a} -> {b}")
# alone vs overlapped time per family
fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
for s, e, q, n in ks:
    ov = sum(max(0, min(e, e2) - max(s, s2)) for s2, e2, q2, _ in ks if q2 != q and s2 < e and e2 > s)
    f = fam[n]; f[0] += 1; f[1] += (e - s) / 1e3; f[2] += min(ov, e - s) / 1e3
print("family: launches, total us, of which beside another stream's kernel")
for n, (c, tot, ov) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:22]:
    print(f"   {n:46s} {c:3d} {tot:8.0f} {ov:8.0f}")
# full listing (optional third argument "list"): every kernel of the step in start order — start, duration, stream, name
if len(sys.argv) > 3 and sys.argv[3] == "list":
    print("listing: start us, duration us, stream, kernel")
    for s, e, q, n in ks:
        print(f"   {s / 1e3:8.1f} {(e - s) / 1e3:7.1f}  s{q}  {n}")
