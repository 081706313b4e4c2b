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
