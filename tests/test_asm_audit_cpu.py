"""Build hygiene of the hand-scheduled LDS-DMA paths (CPU; no kernel is launched).

The LDS-DMA issue statements (csrc/flat_tile.h, sd_common.h, wgrad_gemm.hip, loss_gemm.hip) write M0 without saving it, drop
the s_nop 4 that covers a VALU-written scalar base, and — clip_dz — count their own vmcnt: all of that is sound only while the
COMPILED code has the properties tools/check_dma_hazard.py checks.  A hipcc upgrade or an unrelated edit that changes register
allocation would turn them into silent wrong-data bugs, so the audit runs here, on the code objects of the built library (what
ships) and on a fresh `hipcc -S` listing of loss_gemm.hip (the asm VGPR loads need the statement markers of a listing)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_lds_dma_statements_of_the_built_library(tmp_path):
    import check_dma_hazard as H
    so = os.path.join(ROOT, "speech_decoding_amd", "libsdamd.so")
    if not (os.path.exists(os.path.join(LLVM, "llvm-objdump")) and os.path.exists(so)):
        pytest.skip("needs the ROCm LLVM tools and the built library")
    work = tmp_path / "co"
    work.mkdir()
    shutil.copy(so, work / "libsdamd.so")
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "libsdamd.so"], cwd=work, check=True, capture_output=True)
    objs = [f for f in os.listdir(work) if "gfx950" in f]
    assert objs
    total = 0
    for f in objs:
        dis = work / (f + ".dis")
        with open(dis, "w") as out:
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f], cwd=work, check=True, stdout=out)
        hazards, m0_bad, ndma = H.audit_disassembly(str(dis))
        assert hazards == 0, f"{f}: VALU-written scalar base feeds an LDS-DMA inside 5 wait states"
        assert m0_bad == 0, f"{f}: M0 is used outside the LDS-DMA statements (they do not save it)"
        total += ndma
    assert total > 2000, total          # conv / wgrad / loss / similarity kernels: a few thousand issue sites


def test_asm_loads_of_clip_dz_are_not_touched_before_their_wait(tmp_path):
    import check_dma_hazard as H
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("needs hipcc")
    lst = tmp_path / "loss_gemm.s"
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only", "-o", str(lst),
                    os.path.join(ROOT, "speech_decoding_amd", "csrc", "loss_gemm.hip")], check=True, capture_output=True)
    text = open(lst).read()
    assert text.count("ASMSTART") > 100
    assert H.main(str(lst)) == 0
