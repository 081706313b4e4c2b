"""Static instruction mix of the innermost MFMA loops of every kernel in a hipcc -S listing (diagnostic):
MFMA / LDS-DMA / SALU / VALU / LDS instructions per loop body, and compiler-generated `s_waitcnt vmcnt` inside it
(a compiler wait counts our inline-asm LDS-DMA as well: inside a K loop it drains the DMA pipeline).
    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o k.s k.hip && python tools/loop_mix.py k.s [name filter]"""
import re, subprocess, sys


def main(path, flt="unsigned short"):
    src = open(path).read().split("\n")
    starts = [(i, l.split(":")[0]) for i, l in enumerate(src) if re.match(r"^_Z\S+:", l)]
    starts.append((len(src), "end"))
    for (a, name), (b, _) in zip(starts, starts[1:]):
        body = src[a:b]
        if not any("v_mfma" in l for l in body):
            continue
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = dem.replace("sda::(anonymous namespace)::", "").replace("sda::", "")
        if flt not in dem:
            continue
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        has = lambda s, e: any("v_mfma" in body[k] for k in range(s, e))
        inner = [(s, e) for s, e in loops if has(s, e) and not any(s2 >= s and e2 <= e and (s2, e2) != (s, e) and has(s2, e2) for s2, e2 in loops)]
        out = []
        for s, e in inner:
            inasm = False
            c = dict(mfma=0, dma=0, salu=0, valu=0, lds=0, cwait=0)
            for l in body[s:e]:
                t = l.strip()
                if "#ASMSTART" in t: inasm = True; continue
                if "#ASMEND" in t: inasm = False; continue
                if not t or t.startswith(";") or t.startswith("."):
                    continue
                op = t.split()[0]
                if op == "s_waitcnt" and "vmcnt" in t and not inasm: c["cwait"] += 1
                if op.startswith("v_mfma"): c["mfma"] += 1
                elif op.startswith("global_load_lds"): c["dma"] += 1
                elif op.startswith("s_"): c["salu"] += 1
                elif op.startswith("v_"): c["valu"] += 1
                elif op.startswith("ds_"): c["lds"] += 1
            if c["mfma"] >= 8:
                out.append(c)
        print(dem[:78])
        for c in out:
            print(f"      mfma {c['mfma']:4d}  dma {c['dma']:3d}  salu {c['salu']:4d} ({c['salu'] / c['mfma']:.2f}/mfma)  valu {c['valu']:4d} "
                  f"({c['valu'] / c['mfma']:.2f}/mfma)  lds {c['lds']:3d}  compiler vmcnt waits {c['cwait']}")


if __name__ == "__main__":
    main(*sys.argv[1:])
