"""Diagnostic: a few launches of sim_gemm alone at one shape (for rocprofv3 --pmc passes): python sim_one.py Bm Bn [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from speech_decoding_amd import ops, lib as L
lib = L.load()
Bm, Bn = int(sys.argv[1]), int(sys.argv[2]); n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = "cuda:0"; T, F = 360, 1024
Yt = ops.new_rows(Bm, T, F, torch.bfloat16, dev); Zt = ops.new_rows(Bn, T, F, torch.bfloat16, dev)
ops.rows_view(Yt, Bm, F, T).normal_(); ops.rows_view(Zt, Bn, F, T).normal_()
K = L.rows_tp(T) * F; Np = L.pad_channels(Bn)
ks = lib.sda_sim_gemm_ksplit(Bm, Bn, K, 1)
partial = torch.empty((ks, Bm, Np), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(n):
    L.check(lib.sda_sim_gemm(Yt.data_ptr(), Zt.data_ptr(), partial.data_ptr(), Bm, Bn, Np, K, K, ks, 1, st), "sim_gemm")
torch.cuda.synchronize()
