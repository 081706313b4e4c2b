"""Run one conv_gemm / wgrad_gemm configuration a few times (for rocprofv3 counter passes; diagnostic)."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from speech_decoding_amd import ops

def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "conv"
    cin, cout, KS, dil = (int(v) for v in (sys.argv[2:6] if len(sys.argv) > 5 else (320, 320, 3, 4)))
    flags = int(sys.argv[6]) if len(sys.argv) > 6 else 0
    n = int(os.environ.get("N", 20))
    dev, B, T, dtype = "cuda:0", 256, 360, (torch.float32 if os.environ.get("DTYPE") == "fp32" else torch.bfloat16)
    x = ops.new_rows(B, T, cin, dtype, dev); x.normal_()
    w = torch.randn(cout, cin, KS, device=dev) / math.sqrt(KS * cin)
    wp = ops.pack_conv_weight(w, cout, cin, dtype)
    y = ops.new_rows(B, T, cout, dtype, dev)
    bias = torch.zeros(cout, device=dev)
    stats = torch.zeros((ops.conv_stats_rows(B, T, KS, cout, flags), 2, cout), device=dev)
    res = x if cin == cout else None
    if what == "conv":
        for _ in range(n):
            ops.conv_gemm(x, wp, y, B=B, T=T, KS=KS, dil=dil, bias=bias, res=res, stats=stats, flags=flags)
    else:
        dy = ops.new_rows(B, T, cout, dtype, dev); dy.normal_()
        nseg = int(os.environ.get("NSEG", 24))
        seg = torch.from_numpy(np.floor(np.linspace(0, B, nseg + 1)).astype(np.int32)).to(dev)
        perm = torch.arange(B, dtype=torch.int32, device=dev)
        for _ in range(n):
            ops.wgrad_gemm(dy, x, B=B, T=T, KS=KS, dil=dil, perm=perm, seg_start=seg, nseg=nseg)
    torch.cuda.synchronize()

if __name__ == "__main__":
    main()
