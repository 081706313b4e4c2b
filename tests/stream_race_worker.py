#!/usr/bin/env python3
"""Worker of tests/test_stream_race_gpu.py — run in a FRESH process so that every buffer of the first run is first-use
(a new row-layout buffer is zero-filled on the stream that is current when it is created: the class of bug behind commit
2ef6d75 is a side stream touching such a buffer before that fill has been ordered in front of it).

    python tests/stream_race_worker.py <fp32|bf16|fp16>

Two training steps (with an Adam update between them, so step 2 re-packs changed weights on the packing stream) of a
config-2-shaped model (208 sensors, 27 subjects, full layer widths, 16 segments), four times:
    every side stream ON  / everything on ONE stream,  single process  and  data-parallel path at world size 1 (RCCL)
and within each pair every output — embeddings, loss, ranks, every gradient, of both steps — must be BITWISE equal: the
kernels and their summation orders are identical, only the streams they are queued on differ.  No host synchronisation
happens inside a run (a synchronisation hides a missing event).  Exit code 0 and "stream race check ok" on success."""
import datetime
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import brain_oracle as O                                   # noqa: E402  (seeded parameters / inputs only)
from tests.test_dp_gpu import REAL_208, build                          # noqa: E402
from speech_decoding_amd import loss as sda_loss                       # noqa: E402
from speech_decoding_amd.distributed import active_group, allreduce_gradients   # noqa: E402
from speech_decoding_amd.optim import FusedAdam                        # noqa: E402

DEV = "cuda:0"


def run(dtype, d, P, X, Y, subj, streams: bool, dp: bool):
    enc, lossf = build(dtype, P, d, DEV)
    enc.sync_batchnorm = dp
    lossf.global_negatives = dp
    e = enc.engine
    assert (e.group is not None) == dp
    e.wgrad_side_stream = e.pack_on_side_stream = e.bias_sums_on_side = streams
    sda_loss.PREFETCH_ON_SIDE_STREAM = streams
    params = list(enc.parameters()) + list(lossf.parameters())
    opt = FusedAdam(params, lr=3e-4)
    Xd, Yd = X.to(DEV), Y.to(DEV)
    torch.cuda.synchronize()                                           # inputs are there; from here on: no host sync
    out = []
    for step in range(2):
        np.random.seed(5 + step)                                       # the dropout centre of this step
        lossf.prefetch(Yd, enc.compute_dtype)
        Z = enc(Xd, subj)
        loss = lossf(Yd, Z)
        cnt = sda_loss.retrieval_ranks(Yd, Z)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if dp:
            assert enc.grads_are_reduced
            allreduce_gradients(list(lossf.parameters()))
        rec = {"Z": Z.detach().clone(), "loss": loss.detach().clone(), "ranks": cnt.clone()}
        for n, p in list(enc.named_parameters()) + [("temp", lossf.temp)]:
            rec["grad " + n] = (torch.view_as_real(p.grad) if p.grad.is_complex() else p.grad).detach().clone()
        out.append(rec)
        opt.step()
    rec = {"param " + n: (torch.view_as_real(p) if p.is_complex() else p).detach().clone() for n, p in enc.named_parameters()}
    for n, b in enc.named_buffers():
        if "running" in n:
            rec["buffer " + n] = b.detach().clone()
    out.append(rec)
    torch.cuda.synchronize()
    return out


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "fp32"
    torch.cuda.set_device(DEV)
    os.environ.setdefault("SDA_DP_SINGLE_RANK", "1")
    dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120), device_id=torch.device(DEV))
    assert active_group() is not None
    d = dict(REAL_208, B=16)
    loc = O.synthetic_positions(d["C"], seed=1)
    P = O.seeded_params(d["C"], d["S"], d["D1"], d["D2"], d["F"], d["K"], seed=2, loc=loc)
    X, Y, subj = O.synthetic_batch(d["B"], d["C"], d["T"], d["F"], d["S"], seed=3)
    bad = []
    for dp in (False, True):
        multi = run(dtype, d, P, X, Y, subj, True, dp)                 # FIRST: the run whose buffers are all first-use
        single = run(dtype, d, P, X, Y, subj, False, dp)
        for i, (a, b) in enumerate(zip(multi, single)):
            assert a.keys() == b.keys()
            for k in a:
                if not torch.equal(a[k], b[k]):
                    diff = (a[k].double() - b[k].double()).abs().max()
                    bad.append(f"dp={dp} step/record {i} {k}: max |difference| {float(diff):.3e}")
    dist.barrier()
    dist.destroy_process_group()
    if bad:
        print("STREAM RACE: outputs differ between the multi-stream and the single-stream schedule:\n  " + "\n  ".join(bad[:40]))
        sys.exit(1)
    print("stream race check ok")


if __name__ == "__main__":
    main()
