"""Can a K loop (LDS-DMA + MFMA) and an epilogue (HBM stores) overlap on the same CUs?  (diagnostic)
conv3_flat built K-loop-only (flag 256) and epilogue-only (flag 512), one workgroup per CU each, alone and on two streams."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech_decoding_amd import ops, lib as L

dev, dtype = "cuda:0", torch.bfloat16
B, T, C = 256, 360, 320
x = ops.new_rows(B, T, C, dtype, dev); x.normal_()
w = torch.randn(C, C, 3, device=dev) / math.sqrt(3 * C)
wp = ops.pack_conv_weight(w, C, C, dtype)
bias = torch.zeros(C, device=dev)
FLAT, ONE = L.CONV_FLAT_TILES, L.CONV_ONE_PER_CU
def mk(flags):
    y = ops.new_rows(B, T, C, dtype, dev)
    st = torch.zeros((ops.conv_stats_rows(B, T, 3, C, FLAT), 2, C), device=dev)
    return lambda: ops.conv_gemm(x, wp, y, B=B, T=T, KS=3, dil=4, bias=bias, res=x, stats=st, flags=flags)
N = 10
def wall(fns_by_stream):
    for s, f in fns_by_stream:
        with torch.cuda.stream(s):
            f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main = torch.cuda.current_stream()
    e0.record(main)
    for s, _ in fns_by_stream: s.wait_event(e0)
    for s, f in fns_by_stream:
        with torch.cuda.stream(s):
            for _ in range(N): f()
    for s, _ in fns_by_stream:
        ev = torch.cuda.Event(); ev.record(s); main.wait_event(ev)
    e1.record(main); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / N * 1e3
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
full2, full1 = mk(FLAT), mk(FLAT | ONE)
k1, e1_, k2, e2_ = mk(FLAT | ONE | 256), mk(FLAT | ONE | 512), mk(FLAT | 256), mk(FLAT | 512)
for rep in range(2):
    print(f"full kernel, two workgroups per CU      {wall([(s1, full2)]):7.1f} us")
    print(f"full kernel, one workgroup per CU       {wall([(s1, full1)]):7.1f} us")
    print(f"K loop only, two per CU                 {wall([(s1, k2)]):7.1f} us")
    print(f"epilogue only, two per CU               {wall([(s1, e2_)]):7.1f} us")
    print(f"K loop only, one per CU                 {wall([(s1, k1)]):7.1f} us")
    print(f"epilogue only, one per CU               {wall([(s1, e1_)]):7.1f} us")
    print(f"K-only (1/CU) || epilogue-only (1/CU)   {wall([(s1, k1), (s2, e1_)]):7.1f} us")
    print(f"K-only (1/CU) || K-only (1/CU)          {wall([(s1, k1), (s2, mk(FLAT | ONE | 256))]):7.1f} us")
    print(f"epi-only (1/CU) || epi-only (1/CU)      {wall([(s1, e1_), (s2, mk(FLAT | ONE | 512))]):7.1f} us")
    print(f"full (1/CU) || full (1/CU)              {wall([(s1, full1), (s2, mk(FLAT | ONE))]):7.1f} us", flush=True)
