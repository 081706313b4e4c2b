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
        if op.startswith("global_load") and "s[" in t:
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
    bad += early_touch(lines)
    return bad


def early_touch(lines):
    """An inline-asm load into VGPRs (global_load_dword* inside ;;#ASMSTART ... ;;#ASMEND) is invisible to hipcc's wait
    bookkeeping: its destination counts as written when the statement ends.  Nothing may read or write those registers before
    the next `s_waitcnt vmcnt` — a compiler copy or spill in between moves garbage.  Linear scan in layout order."""
    bad, inasm, pending = 0, False, []          # pending: (set of vgpr numbers, text)
    kernel = ""
    for ln in lines:
        m = re.match(r"^(_Z\S+):", ln)
        if m:
            kernel, pending = m.group(1), []
            continue
        t = ln.strip()
        if "#ASMSTART" in t:
            inasm = True
            continue
        if "#ASMEND" in t:
            inasm = False
            continue
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if op == "s_waitcnt" and "vmcnt" in t:
            pending = []
            continue
        if inasm and re.match(r"global_load_dword(x[234])?$", op):
            mm = re.match(r"\S+\s+v\[(\d+):(\d+)\]", t) or re.match(r"\S+\s+v(\d+),", t)
            if mm:
                lo = int(mm.group(1)); hi = int(mm.group(2)) if mm.lastindex and mm.lastindex > 1 else lo
                pending.append((set(range(lo, hi + 1)), t))
            continue
        if pending:
            regs = set()
            for a, b in re.findall(r"v\[(\d+):(\d+)\]", t):
                regs |= set(range(int(a), int(b) + 1))
            regs |= {int(a) for a in re.findall(r"(?<![\w\[])v(\d+)\b", t)}
            for dst, txt in pending:
                if regs & dst:
                    print(f"EARLY TOUCH in {kernel[:70]}: '{t}' uses the destination of '{txt}' before any vmcnt wait")
                    bad += 1
                    pending = [p_ for p_ in pending if p_[1] != txt]
                    break
    print(f"  {bad} touches of an asm load's destination before its wait")
    return bad


def audit_disassembly(path, quiet=False):
    """The same two properties on `llvm-objdump -d` output of a BUILT gfx950 code object (what actually ships; wired into
    tests/test_asm_audit_cpu.py next to the register-spill guard):
      * no global_load_lds whose scalar base pair was written by a VALU instruction inside the 5 wait states before it;
      * M0 belongs to the LDS-DMA statements alone: the asm statements write it and do not save it, which is only sound while
        hipcc keeps no value of its own there — every instruction that names m0 must be an `s_mov_b32 m0, <scalar>` whose next
        instruction but s_nop is a global_load_lds (the statement's own pattern); anything else (a read of m0, an LDS / GWS /
        movrel / sendmsg use, a write that feeds something else) is reported.
    Returns (hazards, m0_violations, lds_dma_count)."""
    hazards = m0_bad = ndma = 0
    kernel, hist = "", []
    pending_m0 = None                                     # text of an `s_mov_b32 m0` still waiting for its global_load_lds
    for ln in open(path):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
        if m:
            if pending_m0:
                m0_bad += 1
                if not quiet: print(f"M0 in {kernel[:80]}: '{pending_m0}' is not followed by an LDS-DMA")
            kernel, hist, pending_m0 = m.group(1), [], None
            continue
        if not ln.startswith("\t"):
            continue
        t = ln.split("//")[0].strip()
        if not t:
            continue
        op = t.split()[0]
        is_dma = op.startswith("global_load_lds")
        if is_dma:
            ndma += 1
            mm = re.search(r"s\[(\d+):(\d+)\]", t)
            if mm:
                regs = {f"s{mm.group(1)}", f"s{mm.group(2)}"}
                wait = 0
                for prev in reversed(hist[-8:]):
                    pop = prev.split()[0]
                    if pop == "s_nop":
                        wait += int(prev.split()[1], 0) + 1
                        continue
                    dst = prev.split()[1].rstrip(",") if len(prev.split()) > 1 else ""
                    if pop.startswith("v_") and dst in regs and wait < 5:
                        hazards += 1
                        if not quiet: print(f"HAZARD in {kernel[:80]}: '{prev}' feeds '{t}' after {wait} wait states")
                    wait += 1
                    if wait >= 5:
                        break
            pending_m0 = None
        elif pending_m0 is not None and op != "s_nop":
            m0_bad += 1
            if not quiet: print(f"M0 in {kernel[:80]}: '{pending_m0}' is followed by '{t}', not by an LDS-DMA")
            pending_m0 = None
        if re.search(r"\bm0\b", t) and not is_dma:
            if re.match(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi|ttmp\d+|0x[0-9a-f]+|\d+)$", t):
                pending_m0 = t
            else:
                m0_bad += 1
                if not quiet: print(f"M0 in {kernel[:80]}: '{t}' uses M0 outside an LDS-DMA statement")
        hist.append(t)
    return hazards, m0_bad, ndma


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--dis":
        h, m, n = audit_disassembly(sys.argv[2])
        print(f"{sys.argv[2]}: {n} LDS-DMA instructions, {h} scalar-base hazards, {m} foreign uses of M0")
        sys.exit(1 if (h or m) else 0)
    sys.exit(1 if main(*sys.argv[1:]) else 0)
