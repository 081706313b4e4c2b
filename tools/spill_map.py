"""Where does a kernel's register-spill code sit?  (diagnostic)  Reads a hipcc -save-temps .s file and, per kernel, counts
scratch loads / stores inside loops that contain MFMAs (the K loop) versus everywhere else (prologue, epilogue).
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -save-temps -c conv3_flat.hip && python tools/spill_map.py conv3_flat-hip-amdgcn-amd-amdhsa-gfx950.s [name filter]"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l.split(":")[0]) for i, l in enumerate(src) if re.match(r"^_Z\S+:", l)]
starts.append((len(src), "end"))
for (a, name), (b, _) in zip(starts, starts[1:]):
    if flt not in name:
        continue
    body = src[a:b]
    mf = [i for i, l in enumerate(body) if "v_mfma" in l]
    sc = [(i, l.strip().split()[0]) for i, l in enumerate(body) if "scratch_" in l]
    if not mf:
        continue
    labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    cnt = {}
    for i, op in sc:
        inner = [(e - s, s, e) for s, e in loops if s <= i <= e and any(s <= x <= e for x in mf)]
        if inner:
            _, s, e = min(inner)
            key = f"inside a loop with {sum(1 for x in mf if s <= x <= e)} MFMAs (lines {s}..{e})"
        else:
            key = "outside every MFMA loop"
        cnt[(key, op)] = cnt.get((key, op), 0) + 1
    print(f"{name[:110]}\n    {len(mf)} MFMAs, {len(sc)} scratch instructions")
    for k, v in sorted(cnt.items()):
        print(f"      {v:4d} x {k[1]:24s} {k[0]}")
