"""Static check of the LDS-DMA issue sequences in a hipcc -S listing (diagnostic / build hygiene).

An SGPR written by a VALU instruction (v_readfirstlane, v_readlane, v_cmp ... to an SGPR) needs 5 wait states before a
VMEM instruction reads it as its address base (cdna ISA, manually inserted wait states).  hipcc pads this for its own
instructions but not for an instruction inside an `asm` statement, so an inline-asm global_load_lds whose scalar base was
just produced by a v_readfirstlane must carry its own s_nop 4.  This script walks every global_load_lds in the listing and
reports the ones whose base pair (or M0 source) was written by a VALU fewer than 5 instructions earlier with no s_nop
covering the gap.  It also counts, per kernel, SALU / VALU / MFMA / LDS-DMA instructions inside loops that contain MFMAs.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S --cuda-device-only -o k.s k.hip && python tools/check_dma_hazard.py k.s
"""
import re
import sys


def main(path, flt=""):
    lines = open(path).read().split("\n")
    bad = 0
    kernel = ""
    hist = []            # (text) of recent real instructions
    for ln in lines:
        m = re.match(r"^(_Z\S+):", ln)
        if m:
            kernel, hist = m.group(1), []
            continue
        t = ln.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "global_load_lds_dwordx4":
            mm = re.search(r"s\[(\d+):(\d+)\]", t)
            if mm:
                regs = {f"s{mm.group(1)}", f"s{mm.group(2)}"}
                wait = 0
                for prev in reversed(hist[-8:]):
                    pop = prev.split()[0]
                    if pop == "s_nop":
                        wait += int(prev.split()[1]) + 1
                        continue
                    dst = prev.split()[1].rstrip(",") if len(prev.split()) > 1 else ""
                    if pop.startswith("v_") and dst in regs and wait < 5:
                        print(f"HAZARD in {kernel[:80]}: '{prev}' feeds '{t}' after {wait} wait states")
                        bad += 1
                    wait += 1
                    if wait >= 5:
                        break
        hist.append(t)
    print(f"{path}: {bad} unpadded VALU -> VMEM scalar-base hazards")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(*sys.argv[1:]) else 0)
