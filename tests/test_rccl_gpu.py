"""RCCL call sequence on ONE MI355X: torch.distributed backend "nccl" (= RCCL) at world size 1 with the data-parallel
code path forced on (SDA_DP_SINGLE_RANK=1).  Every collective of a training step then really goes through RCCL with one
rank — the three communicators (`dist.new_group` for the speech-row gather and the gradient buckets), the asynchronous
all-gather started by `CLIPLoss.prefetch` on a side stream, the SyncBN all-reduces, the mask broadcast, the
`all_reduce(async_op=True)` issued on the weight-gradient side stream with `work.wait()` on the main stream, the
`record_stream` hand-offs — and the result must equal the plain single-process step.  (Two real ranks over xGMI:
tests/test_dp_gpu.py::test_two_ranks_on_rccl_when_two_gpus_are_visible, on a box with two GPUs.)"""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, datetime
    import numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from oracle import brain_oracle as O
    from tests.test_dp_gpu import TOY, REAL_208, build, grads_of
    from speech_decoding_amd.distributed import allreduce_gradients, side_group, active_group
    DEV = "cuda:0"
    torch.cuda.set_device(DEV)
    dist.init_process_group("nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120), device_id=torch.device(DEV))
    assert active_group() is not None
    for d in (TOY, REAL_208):
        loc = O.synthetic_positions(d["C"], seed=1)
        P = O.seeded_params(d["C"], d["S"], d["D1"], d["D2"], d["F"], d["K"], seed=2, loc=loc)
        X, Y, subj = O.synthetic_batch(d["B"], d["C"], d["T"], d["F"], d["S"], seed=3)
        outs = []
        for dp in (True, False):
            enc, lossf = build("fp32", P, d, DEV)
            enc.sync_batchnorm = dp
            lossf.global_negatives = dp
            np.random.seed(5)                                   # same dropout centre; under DP the mask is broadcast
            Yd = Y.to(DEV)
            for step in range(2):                               # second step: communicators and ring buffers reused
                if dp:
                    lossf.prefetch(Yd, torch.float32)
                Z = enc(X.to(DEV), subj)
                loss = lossf(Yd, Z)
                enc.zero_grad(); lossf.zero_grad()
                loss.backward()
                if dp:
                    assert enc.grads_are_reduced and enc.engine.group is not None
                    allreduce_gradients(list(lossf.parameters()))
            torch.cuda.synchronize()
            outs.append((float(loss.detach()), Z.detach().float().cpu(), grads_of(enc, lossf)))
        (l1, Z1, g1), (l0, Z0, g0) = outs
        # the DP path sums the BatchNorm partials in a different (fp32 slab) order: equal to fp32 rounding, not bitwise
        assert abs(l1 - l0) < 2e-5 * max(1.0, abs(l0)), (l1, l0)
        assert float((Z1 - Z0).abs().max()) <= 1e-4 * float(Z0.abs().max()), float((Z1 - Z0).abs().max()) / float(Z0.abs().max())
        for k in g0:
            if k.startswith("conv_blocks.") and k.endswith((".conv0.bias", ".conv1.bias")):
                continue                                        # null gradients: rounding noise on both sides
            assert float((g1[k] - g0[k]).abs().max()) <= 2e-3 * float(g0[k].abs().max()) + 1e-7, k
        assert side_group("grads", active_group()) is not active_group()        # communicators of their own exist
    # ---- bench.py --emulate-world: one rank behaving as rank 0 of W (distributed.emulate_world): the loss sees W * B speech
    # rows — its own through the real all-gather, (W - 1) * B resident stand-ins — and must equal the engine's block form on
    # exactly those rows (the tiled embedding gradient included: 288 speech rows x 96 columns, bf16)
    from speech_decoding_amd import CLIPLoss, engine as E, lib as L, ops
    from speech_decoding_amd import loss as sda_loss
    from speech_decoding_amd.distributed import emulate_world
    class A(dict):
        __getattr__ = dict.__getitem__
    B, W, F, T = 96, 3, 64, 44
    g = torch.Generator().manual_seed(31)
    Yl, Zl = torch.randn(B, F, T, generator=g).to(DEV), torch.randn(B, F, T, generator=g).to(DEV)
    emulate_world(W)
    lossf = CLIPLoss(A(reduction="mean", init_temperature=5.1)).to(DEV)
    Zt = sda_loss.as_rows(Zl, B, F, T, torch.bfloat16, "Z")
    Zv = ops.rows_view(Zt, B, F, T).requires_grad_(True)
    lossf.prefetch(Yl, torch.bfloat16)
    loss = lossf(Yl, Zv)
    loss.backward()
    rows, nsq = next(iter(lossf._state.emulated.values()))
    Tp = L.rows_tp(T)
    Yall = ops.new_rows(W * B, T, F, torch.bfloat16, DEV)
    ops.pack_rows(Yl, Yall)
    Yall[B * Tp: W * B * Tp].copy_(rows)
    share, logits, cnt, cctx = E.clip_forward(Yall, Zt, lossf.temp.detach(), Bm=W * B, Bn=B, T=T, col0=0, B_global=W * B)
    dZt = ops.new_rows(B, T, F, torch.bfloat16, DEV)
    E.clip_backward(cctx, dZt, torch.ones(1, device=DEV))
    torch.cuda.synchronize()
    assert abs(float(loss) - float(share)) <= 1e-6 * max(1.0, abs(float(share))), (float(loss), float(share))
    assert torch.equal(lossf.last_logits, logits) and tuple(logits.shape) == (W * B, B)
    assert torch.equal(Zv.grad, ops.rows_view(dZt, B, F, T))
    assert abs(float(lossf.temp.grad) - float(cctx.dtemp)) <= 1e-6 * max(1.0, abs(float(cctx.dtemp)))
    emulate_world(1)
    dist.barrier()
    dist.destroy_process_group()
    print("rccl single-rank ok")
""")


def test_data_parallel_step_through_rccl_at_world_size_one():
    with __import__("socket").socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SDA_DP_SINGLE_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", WORKER % ROOT], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    log_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(log_dir):
        open(os.path.join(log_dir, "rccl_worker.log"), "w").write(out.stdout + "\n---- stderr ----\n" + out.stderr)
    tail = "\n".join(l for l in out.stderr.splitlines() if "Error" in l or "error" in l or "assert" in l or "line " in l)[-3000:]
    assert out.returncode == 0 and "rccl single-rank ok" in out.stdout, tail
