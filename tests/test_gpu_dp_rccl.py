"""Data parallel on real GPUs: two ranks over RCCL (backend 'nccl'), one GPU each - the GPU
twin of tests/test_dp_gloo.py.  `Trainer.train_step` on 2 x B/2 must equal 1 x B: same loss
(mean of the ranks' means), same parameters after the Adam step.  Skipped unless the box has at
least two GPUs (the 1-GPU boxes of the build round do not)."""

import os
import socket
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def _setup(batch=4, frames=61, seed=5):
    from ctc_asr_amd.model import ModelConfig, init_params
    cfg = ModelConfig(used_model='ds2', conv_filters=(4, 4), num_units_dense=32,
                      num_layers_rnn=2, num_units_rnn=64, rnn_cell='lstm', cudnn=True,
                      dense_dropout_rate=0.0)
    rng = np.random.default_rng(seed)
    flat = init_params(cfg, seed)
    for name in flat:
        flat[name] = (flat[name] + rng.normal(size=flat[name].shape) * 0.05).astype(np.float32)
    feats = rng.normal(size=(batch, frames, 80)).astype(np.float32)
    flen = np.full(batch, frames, dtype=np.int32)
    labels = [list(rng.integers(1, 28, size=rng.integers(1, 9))) for _ in range(batch)]
    return cfg, flat, feats, flen, labels


def _worker(rank, world, port, out, backend='nccl', share_gpu=False):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from ctc_asr_amd.engine import Trainer, init_distributed
    device = 0 if share_gpu else rank
    if share_gpu:
        os.environ['LOCAL_RANK'] = '0'
    init_distributed(backend)
    torch.cuda.set_device(device)
    cfg, flat, feats, flen, labels = _setup()
    half = len(labels) // world
    lo, hi = rank * half, (rank + 1) * half
    # tiny buckets: several all-reduces per step, launched from the backward hooks
    trainer = Trainer(cfg, device='cuda:{}'.format(device), params=flat, world_size=world,
                      rank=rank, bucket_bytes=4096)
    trainer.lr = 1e-3
    loss = trainer.train_step(torch.tensor(feats[lo:hi]), torch.tensor(flen[lo:hi]),
                              labels[lo:hi])
    mean_loss = float(trainer.global_mean(loss))
    params = trainer.model.arena.param.cpu().numpy()
    launched = trainer.reducer.launched
    gathered = [None] * world
    dist.all_gather_object(gathered, params)
    if rank == 0:
        out.put((mean_loss, gathered, launched, dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


def _two_ranks_against_one(backend, share_gpu):
    import torch.multiprocessing as mp
    from ctc_asr_amd.engine import Trainer
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(rank, 2, port, out, backend, share_gpu))
             for rank in range(2)]
    for proc in procs:
        proc.start()
    # read the result BEFORE joining: rank 0 blocks in put() until the pipe is drained (the
    # parameters are larger than the pipe buffer), and rank 1 waits for it at the barrier
    deadline = time.time() + 600
    while out.empty():
        assert time.time() < deadline and all(p.exitcode in (None, 0) for p in procs), \
            [p.exitcode for p in procs]
        time.sleep(0.2)
    mean_loss, gathered, launched, used = out.get()
    for proc in procs:
        proc.join(120)
        assert proc.exitcode == 0
    assert used == backend and launched >= 2
    assert np.array_equal(gathered[0], gathered[1])         # replicas stay identical
    cfg, flat, feats, flen, labels = _setup()
    single = Trainer(cfg, device='cuda:0', params=flat)
    single.lr = 1e-3
    loss = single.train_step(torch.tensor(feats), torch.tensor(flen), labels)
    assert abs(float(loss) - mean_loss) < 1e-5
    want = single.model.arena.param.cpu().numpy()
    assert np.abs(gathered[0] - want).max() < 1e-6


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (RCCL over xGMI)')
def test_two_rccl_ranks_equal_one_rank_full_batch():
    _two_ranks_against_one('nccl', share_gpu=False)


@pytest.mark.parametrize('early', ['0', '1'])
def test_two_gloo_ranks_sharing_one_gpu_equal_one_rank_full_batch(early, monkeypatch):
    """The same check where only one GPU exists: both ranks on cuda:0, collectives through gloo.
    Everything but the transport is the production path: gradient hooks, buckets launched from
    the main and the side stream, the held-back release (default) or the per-layer early
    release (CTCASR_ALLREDUCE_EARLY=1), Adam's 1 / world scaling."""
    monkeypatch.setenv('CTCASR_ALLREDUCE_EARLY', early)      # inherited by the spawned ranks
    _two_ranks_against_one('gloo', share_gpu=True)
