"""Data-parallel logic on CPU: two processes over the gloo backend (the MI355X run uses the same
code with backend 'nccl' = RCCL).  Checks the bucketed gradient reducer and that sharding a
global batch over ranks reproduces the single-process gradient of the batch-mean CTC loss."""

import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ctc_asr_amd.engine import GradientReducer
from oracle import torch_ref
from tests.helpers import make_params


def _free_port():
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def _init(rank, world, port):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)


def _reducer_worker(rank, world, port, out):
    _init(rank, world, port)
    arena = torch.full((1000,), float(rank + 1))
    slices = [('a', 0, 100), ('b', 100, 400), ('c', 400, 1000)]
    for bucket_bytes in (4, 1200, 1 << 20):       # per-slice, merged pairs, one bucket
        arena.fill_(float(rank + 1))
        reducer = GradientReducer(arena, world, bucket_bytes=bucket_bytes)
        for name, start, stop in reversed(slices):    # backward order: end of arena first
            reducer.hook(name, start, stop)
        reducer.finish()
        assert torch.all(arena == 3.0), bucket_bytes
    # hold_until: nothing is launched before the named layer reports, then everything is
    arena.fill_(float(rank + 1))
    reducer = GradientReducer(arena, world, bucket_bytes=4, hold_until='a')
    reducer.hook('c', 400, 1000)
    reducer.hook('b', 100, 400)
    assert not reducer.works and torch.all(arena == float(rank + 1))
    reducer.hook('a', 0, 100)
    assert len(reducer.works) == 1          # below max_bucket_bytes: one merged bucket
    reducer.finish()
    assert torch.all(arena == 3.0)
    # ... and a released range larger than max_bucket_bytes goes out in equal pieces no larger
    # than that (1000 floats, at most 300 per collective -> 4 x 250), every element exactly once
    arena.fill_(float(rank + 1))
    reducer = GradientReducer(arena, world, bucket_bytes=4, hold_until='a',
                              max_bucket_bytes=1200)
    for name, start, stop in reversed(slices):
        reducer.hook(name, start, stop)
    assert len(reducer.works) == 4 and reducer.launched == 4
    # the guard words of a step travel with its gradients: their maximum over ranks comes back,
    # so every replica drops a step that ONE of them must not apply
    guard = torch.tensor([1 if rank == 1 else 0, 5 * rank], dtype=torch.int32)
    reducer.finish(guard)
    assert guard.tolist() == [1, 5 * (world - 1)]
    assert torch.all(arena == 3.0)
    # force=True runs the collectives whatever the world size; inactive = no-op
    idle = GradientReducer(arena, 1, bucket_bytes=4)
    idle.hook('c', 400, 1000)
    idle.finish()
    assert idle.launched == 0
    if rank == 0:
        out.put('ok')
    dist.destroy_process_group()


def _grad_vector(model):
    return torch.cat([p.grad.reshape(-1) for p in model.parameters()])


def _shard_worker(rank, world, port, out):
    _init(rank, world, port)
    rng = np.random.default_rng(0)
    params = make_params(rng, 'ds2', 'lstm', hidden=8, dense=12, conv_filters=(4, 4))
    feats = rng.normal(size=(4, 41, 80))
    lengths = np.full(4, 41)
    labels = [[1, 2, 3], [4, 4], [5], [6, 7, 8, 9]]
    model = torch_ref.TorchRefModel(params, 'ds2', 'lstm', True, dtype=torch.float64)
    lo, hi = rank * 2, rank * 2 + 2
    logits, seq_len = model(torch.tensor(feats[lo:hi]), lengths[lo:hi])
    loss, _ = model.loss(logits, seq_len, labels[lo:hi])
    loss.backward()
    arena = _grad_vector(model).clone()
    reducer = GradientReducer(arena, world, bucket_bytes=4096)
    third = arena.numel() // 3
    for name, start, stop in (('top', 2 * third, arena.numel()), ('mid', third, 2 * third),
                              ('front', 0, third)):
        reducer.hook(name, start, stop)
    reducer.finish()
    arena /= world
    if rank == 0:
        full = torch_ref.TorchRefModel(params, 'ds2', 'lstm', True, dtype=torch.float64)
        logits, seq_len = full(torch.tensor(feats), lengths)
        loss, _ = full.loss(logits, seq_len, labels)
        loss.backward()
        out.put(float((arena - _grad_vector(full)).abs().max()))
    dist.destroy_process_group()


def _run(worker):
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=worker, args=(rank, 2, port, out)) for rank in range(2)]
    for proc in procs:
        proc.start()
    for proc in procs:
        proc.join(120)
        assert proc.exitcode == 0
    return out.get()


def test_bucketed_reducer_sums_every_slice():
    assert _run(_reducer_worker) == 'ok'


def test_two_ranks_times_half_batch_equals_one_rank_full_batch():
    assert _run(_shard_worker) < 1e-12
