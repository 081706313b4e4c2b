"""Diagnostic: the parameter-space products of the composed SubjectBlock's backward (the serial tail of a training step) alone,
at config-2 shapes (S = 27 subjects, D1 = 270, D2 = 320, C = 208): microseconds per launch, checked against fp64 once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech_decoding_amd import ops
dev = "cuda:0"
S, D1, D2p, C, Cp = 27, 270, 320, 208, 256
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s: torch.randn(*s, device=dev, generator=g)
W0cat = r(D1, 3 * D2p); M = r(S, 3 * D2p, Cp); T1aug = r(D1, C + 1); Ws = r(S, D1, D1); Wd = r(D1, C); sbw = r(D1, D1)
G = ops.param_gemm(W0cat, M[:, :, : C + 1])
dT1f = r(D1, C + 1); dT1 = dT1f[:, :C]
cases = [("G = W0cat . M[s]            (27 x 270 x 960 x 209)", lambda: ops.param_gemm(W0cat, M[:, :, : C + 1])),
         ("subj_w = G[s] . T1aug^T     (27 x 270 x 209 x 270)", lambda: ops.param_gemm(G, T1aug.t())),
         ("part = W_subj[s]^T . G[s]   (27 x 270 x 270 x 209)", lambda: ops.param_gemm(Ws.transpose(1, 2), G)),
         ("sb_w = dT1 . Wd^T           (270 x 208 x 270)", lambda: ops.param_gemm(dT1, Wd.t())),
         ("dWd = sb_w^T . dT1          (270 x 270 x 208)", lambda: ops.param_gemm(sbw.t(), dT1))]
ref = (W0cat.double() @ M[:, :, : C + 1].double())
print("G vs fp64: max rel", float((G.double() - ref).abs().max() / ref.abs().max()))
for name, fn in cases:
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): fn()
    e1.record(); torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) * 20:.1f} us")
