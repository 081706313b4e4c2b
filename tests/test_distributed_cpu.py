"""world_size-2 `gloo` tests (CPU) of the data-parallel host logic: gradient bucket all-reduce, parameter
broadcast, the cross-rank softmax-statistic merge used for global-batch negatives, and the algebra that
makes per-rank loss shares / gradients sum to the single-process result (checked with the oracle)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import brain_oracle as O


def _worker(rank, world, port, fn, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    # (the package's own entry point: the RCCL stream-priority option of backend "nccl" does not apply to gloo, the call must
    # pass everything else through)
    from speech_decoding_amd.distributed import init_process_group
    init_process_group("gloo", rank=rank, world_size=world)
    try:
        ret[rank] = fn(rank, world)
    finally:
        dist.destroy_process_group()


def _free_port():
    """A port the OS hands out (a pid-derived one can collide with another test process)."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def run2(fn):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, fn, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    return dict(ret)


def _grads(rank, world):
    from speech_decoding_amd.distributed import allreduce_gradients, broadcast_parameters
    torch.manual_seed(rank)
    lin = torch.nn.Linear(5, 3)
    z = torch.nn.Parameter(torch.randn(4, 2, dtype=torch.cfloat))
    absent = torch.nn.Parameter(torch.randn(7))
    broadcast_parameters(lin)
    w0 = lin.weight.detach().clone()
    lin.weight.grad = torch.full_like(lin.weight, float(rank + 1))
    lin.bias.grad = torch.full_like(lin.bias, 10.0 * (rank + 1))
    z.grad = torch.full_like(z, complex(rank + 1, -(rank + 1)))
    if rank == 0:
        absent.grad = torch.ones(7)
    allreduce_gradients([lin.weight, lin.bias, z, absent], bucket_bytes=64)
    return (w0, lin.weight.grad.clone(), lin.bias.grad.clone(), z.grad.clone(), absent.grad.clone())


def test_gradient_allreduce_and_broadcast():
    out = run2(_grads)
    assert torch.equal(out[0][0], out[1][0])                         # broadcast made the replicas equal
    for r in (0, 1):
        assert torch.all(out[r][1] == 3.0) and torch.all(out[r][2] == 30.0)
        assert torch.all(out[r][3] == complex(3, -3))
        assert torch.all(out[r][4] == 1.0)                           # None grad on one rank counts as zero


def _clip_shares(rank, world):
    """Global-negative CLIP loss from per-rank column blocks == single-process loss (oracle logits)."""
    from speech_decoding_amd.distributed import merge_row_softmax_stats, shard_range
    g = torch.Generator().manual_seed(0)
    Bg, D = 12, 40
    Y, Z = torch.randn(Bg, D, 1, generator=g), torch.randn(Bg, D, 1, generator=g)
    temp = torch.tensor([1.3])
    logits = O.clip_logits(Y, Z, temp)                               # (Bg, Bg) rows = speech
    lo, hi = shard_range(Bg, rank, world)
    blk = logits[:, lo:hi]                                           # this rank's columns
    row_max = blk.max(dim=1).values
    row_sum = torch.exp(blk - row_max[:, None]).sum(dim=1)
    owned = torch.zeros(Bg)
    owned[lo:hi] = blk[lo:hi].diag()                                 # positives whose column lives on this rank
    row_lse, gdiag = merge_row_softmax_stats(row_max, row_sum, dist.group.WORLD, diag=owned)
    assert torch.allclose(gdiag, logits.diag(), atol=1e-6)           # packed into the same collective
    assert torch.allclose(row_lse, merge_row_softmax_stats(row_max, row_sum, dist.group.WORLD), atol=1e-6)
    col_lse = torch.logsumexp(blk, dim=0)
    diag = blk[lo:hi].diag()
    share = ((row_lse[lo:hi] - diag) + (col_lse - diag)).sum() / (2 * Bg)
    tot = share.clone()
    dist.all_reduce(tot)
    full, _ = O.clip_loss(Y, Z, temp)
    return float(tot), float(full), bool(torch.allclose(row_lse, torch.logsumexp(logits, dim=1), atol=1e-5))


def test_global_negative_loss_shares_sum_to_full_loss():
    out = run2(_clip_shares)
    for r in (0, 1):
        tot, full, lse_ok = out[r]
        assert lse_ok and abs(tot - full) < 1e-5


def _syncbn(rank, world):
    """Summing per-rank (sum, sumsq) partials reproduces the global-batch BatchNorm of the oracle."""
    g = torch.Generator().manual_seed(1)
    x = torch.randn(8, 6, 10, generator=g)
    lo, hi = rank * 4, rank * 4 + 4
    part = torch.stack([x[lo:hi].sum(dim=(0, 2)), (x[lo:hi] ** 2).sum(dim=(0, 2))])
    dist.all_reduce(part)
    n = 8 * 10
    mean = part[0] / n
    var = part[1] / n - mean ** 2
    ref = O.batchnorm_train(x, torch.ones(6), torch.zeros(6), None, "")
    mine = (x - mean[None, :, None]) / torch.sqrt(var[None, :, None] + 1e-5)
    return bool(torch.allclose(ref, mine, atol=1e-5))


def test_synchronised_batchnorm_statistics():
    out = run2(_syncbn)
    assert out[0] and out[1]


def _seed_and_sampler(rank, world):
    """SpatialDropout's centre under data parallelism: NumPy's global generator in lockstep on all ranks
    (distributed.seed_numpy_all_ranks), and the sampler shards of data.ShardedRandomSampler."""
    import numpy as np
    from speech_decoding_amd.data import ShardedRandomSampler
    from speech_decoding_amd.distributed import allreduce_gradients, seed_numpy_all_ranks
    np.random.seed(1000 + rank)                                      # ranks start out of step
    seed = seed_numpy_all_ranks()
    centres = [int(np.random.randint(208)) for _ in range(5)]
    fixed = seed_numpy_all_ranks(77)
    centres2 = [int(np.random.randint(208)) for _ in range(3)]
    mine = [b.tolist() for b in ShardedRandomSampler(50, 8, 3, rank, world, seed=9)]
    # bucketed all-reduce leaves every .grad a view of its bucket (no copy back) with the summed values
    a, b = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2, dtype=torch.cfloat))
    a.grad, b.grad = torch.full((3,), float(rank + 1)), torch.full((2,), complex(rank + 1, 1.0))
    allreduce_gradients([a, b])
    same_storage = a.grad.untyped_storage().data_ptr() == torch.view_as_real(b.grad).untyped_storage().data_ptr()
    return seed, centres, fixed, centres2, mine, a.grad.tolist(), b.grad.tolist(), same_storage


def test_numpy_lockstep_and_sampler_shards():
    from speech_decoding_amd.data import ShardedRandomSampler
    out = run2(_seed_and_sampler)
    assert out[0][0] == out[1][0] and out[0][1] == out[1][1]          # same broadcast seed, same centres afterwards
    assert out[0][2] == out[1][2] == 77 and out[0][3] == out[1][3]
    whole = [b.tolist() for b in ShardedRandomSampler(50, 8, 3, 0, 1, seed=9)]
    assert [x + y for x, y in zip(out[0][4], out[1][4])] == whole
    for r in (0, 1):
        assert out[r][5] == [3.0, 3.0, 3.0] and out[r][6] == [complex(3, 2)] * 2 and out[r][7]
