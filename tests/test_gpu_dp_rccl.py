"""Data parallel on real GPUs: two ranks over RCCL (backend 'nccl'), one GPU each - the GPU
twin of tests/test_dp_gloo.py.  `Trainer.train_step` on 2 x B/2 must equal 1 x B: same loss
(mean of the ranks' means), same parameters after the Adam step.  Skipped unless the box has at
least two GPUs (the 1-GPU boxes of the build round do not).

What a 1-GPU box can do: (1) the same check with both ranks on the one GPU and gloo as the
transport; (2) a world-size-1 `nccl` process group with the reducer forced on - RCCL is loaded
and initialised under HSA_ENABLE_IPC_MODE_LEGACY=0, its communicator and stream plumbing
(ProcessGroupNCCL's own stream, the event hand-offs to and from the main and side streams,
`work.wait()`) run under the real backward pass with the persistent BiLSTM-1024 kernels, in both
release modes; (3) bench.py's world-size-2 path with its dual-mode report."""

import json
import subprocess
import sys

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


def _nccl_world1_worker(port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1',
                      LOCAL_RANK='0', HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch.distributed as dist
    from ctc_asr_amd import hip
    from ctc_asr_amd.engine import Trainer
    from ctc_asr_amd.model import ModelConfig, init_params
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1)
    # a first collective outside the step: communicator set-up (where IPC / topology problems of
    # the box would surface) on its own, then the value check
    probe = torch.arange(8, dtype=torch.float32, device='cuda:0')
    dist.all_reduce(probe)
    torch.cuda.synchronize()
    assert probe.tolist() == list(range(8))
    # DS2 2-conv + 2 x BiLSTM-1024: the persistent recurrence kernels, backward in step ranges
    # with weight-gradient GEMMs on the side stream (T' = 48 -> 3 ranges at batch 20)
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=256,
                      num_layers_rnn=2, num_units_rnn=1024, rnn_cell='lstm', cudnn=True,
                      dense_dropout_rate=0.0)
    batch, frames = 20, 95
    assert hip.rnn_persistent_supported('lstm', 48, batch, 1024)
    rng = np.random.default_rng(3)
    flat = init_params(cfg, 3)
    feats = torch.tensor(rng.normal(size=(batch, frames, 80)).astype(np.float32))
    flen = torch.full((batch,), frames, dtype=torch.int32)
    labels = [list(rng.integers(1, 28, size=rng.integers(3, 12))) for _ in range(batch)]
    results = {}
    for mode in ('off', 'held', 'early'):
        trainer = Trainer(cfg, device='cuda:0', params=flat, world_size=1,
                          bucket_bytes=1 << 20, force_reducer=(mode != 'off'),
                          allreduce_early=(mode == 'early'))
        trainer.lr = 1e-3
        losses = [float(trainer.train_step(feats, flen, labels)) for _ in range(3)]
        trainer.drain_checks()                   # sticky time-out word clean, CTC status clean
        # a one-rank sum is the identity, bit for bit, on the real gradient arena (475 MB at C3,
        # here 50 MB) in one collective
        grad = trainer.model.arena.grad
        summed = grad.clone()
        dist.all_reduce(summed)
        torch.cuda.synchronize()
        assert torch.equal(summed, grad)
        results[mode] = (losses, trainer.model.arena.param.cpu().numpy(),
                         grad.cpu().numpy(), trainer.reducer.launched, trainer.release)
    out.put((results, dist.get_backend()))
    dist.destroy_process_group()


def test_nccl_world_size_one_reducer_forced_on_both_release_modes():
    """RCCL on the box we have: one rank, one GPU, backend 'nccl', `GradientReducer(force=True)`.
    Three training steps per mode must reproduce the reducer-off run - a one-rank sum is the
    identity (checked bit for bit on the gradient arena inside the worker); two runs of the step
    itself agree to fp32 summation order only (the CTC gradient and the bias sums accumulate
    with atomics), hence 1e-6 here - with several bucket launches per step in either release
    mode and no recurrence time-out."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.SimpleQueue()
    proc = ctx.Process(target=_nccl_world1_worker, args=(_free_port(), out))
    proc.start()
    deadline = time.time() + 600
    while out.empty():
        assert time.time() < deadline and proc.exitcode in (None, 0), proc.exitcode
        time.sleep(0.2)
    results, backend = out.get()
    proc.join(120)
    assert proc.exitcode == 0 and backend == 'nccl'
    off = results['off']
    assert off[3] == 0 and np.isfinite(off[0]).all()
    for mode in ('held', 'early'):
        losses, params, grads, launched, release = results[mode]
        assert release == mode
        assert launched >= 3 * 4, (mode, launched)        # buckets of 1 MB: many per step
        assert np.abs(np.array(losses) - np.array(off[0])).max() < 1e-4, (mode, losses, off[0])
        # (Adam divides by sqrt(v): last-bit differences of near-zero gradients move a parameter
        # by a visible fraction of lr = 1e-3 per step)
        assert np.abs(params - off[1]).max() < 1e-4, mode
        assert np.abs(grads - off[2]).max() < 1e-5 * max(1.0, np.abs(off[2]).max()), mode


def test_bench_two_ranks_on_one_gpu_report_both_release_modes():
    """bench.py --gpus 2 as the driver starts it (two processes, torchrun environment), here with
    both ranks on the one GPU and gloo as the transport (tools/two_rank_one_gpu.py): the line
    must carry both release modes, the stubbed step and the per-rank spread."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    run = subprocess.run([sys.executable, os.path.join(root, 'tools', 'two_rank_one_gpu.py'),
                          '--workload', 'tiny', '--steps', '2', '--warmup', '1'],
                         capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line['n_gpus'] == 2 and line['devices_shared'] is True
    modes = line['allreduce']['modes']
    assert set(modes) == {'held', 'early'}
    for mode, info in modes.items():
        assert info['mode'] == mode and info['launches_per_step'] >= 1
        assert len(info['rank_ms_per_step']['all']) == 2
        assert 'exposed_allreduce_ms' in info
    assert line['exposed_allreduce_frac_of_step'] == \
        line['allreduce']['exposed_allreduce_frac_of_step']     # (up front, behind `value`)
    assert list(line).index('exposed_allreduce_frac_of_step') == list(line).index('value') + 1
    assert line['allreduce']['chosen'] in modes
    assert line['allreduce']['stubbed_ms_per_step'] > 0


@pytest.mark.parametrize('traffic', [False, True])
@pytest.mark.parametrize('early', [False, True])
def test_collective_shaped_kernels_beside_the_persistent_recurrences(early, traffic):
    """The first N > 1 run must not be the first time resident, non-yielding workgroups share
    the chip with the persistent recurrences.  NCCL refuses two ranks on one
    device, so the stand-in: every gradient bucket launches a kernel of 24 workgroups that holds
    its CUs for bytes / 150 GB/s on a stream of its own, ordered and waited for exactly like an
    asynchronous all-reduce (`GradientReducer(stand_in=)`), in both release modes - beside the
    128-CU backward recurrences (early) / behind the last of them and in front of the next step's
    whole-chip forward recurrence (held).  DS2 2 x BiLSTM-1024 at batch 32 (the two-tile kernels):
    no recurrence time-out, the losses of the plain run bit for bit (nothing touches the
    gradients), several launches per step; the step-time inflation is printed (DESIGN.md 6 quotes
    bench.py --collective-stand-in for the C3 figure).  ``traffic`` (round 5): the resident
    workgroups also stream 6 x the bucket's bytes through L2 / fabric / HBM meanwhile
    (`ctcasr_collective_traffic`) - a ring all-reduce's memory side beside the recurrences' exchange
    and beside the library's stream-K weight-gradient GEMMs (early mode runs next to both)."""
    from ctc_asr_amd.engine import Trainer
    from ctc_asr_amd.model import CTCModel, ModelConfig
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=2048,
                      num_layers_rnn=2, num_units_rnn=1024, rnn_cell='lstm', cudnn=True,
                      dense_dropout_rate=0.0)
    rng = np.random.default_rng(8)
    feats = torch.tensor(rng.normal(size=(32, 399, 80)).astype(np.float32), device='cuda')
    flen = torch.full((32,), 399, dtype=torch.int32)
    labels = CTCModel.pack_labels([list(rng.integers(1, 28, size=40)) for _ in range(32)], 'cuda')

    def run(stand_in):
        trainer = Trainer(cfg, device='cuda', seed=2, allreduce_early=early,
                          collective_stand_in=stand_in)
        assert trainer.release == ('early' if early else 'held')
        losses = []
        for _ in range(2):
            trainer.train_step(feats, flen, labels)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            losses.append(trainer.train_step(feats, flen, labels))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 6 * 1e3
        trainer.drain_checks()                      # raises on a recurrence time-out
        return torch.stack(losses).cpu().numpy(), ms, trainer.reducer.launched

    plain, plain_ms, launched0 = run(None)
    busy, busy_ms, launched = run((24, 150.0, traffic))
    assert launched0 == 0 and launched >= 8 * 2
    assert np.isfinite(busy).all()
    # (the step itself is deterministic to fp32 summation order only: atomics in the CTC gradient)
    assert np.abs(busy - plain).max() < 1e-4 * np.abs(plain).max()
    print('collective stand-in{}, {} release: {:.2f} -> {:.2f} ms per step ({} launches)'.format(
        ' with memory traffic' if traffic else '', 'early' if early else 'held', plain_ms,
        busy_ms, launched))
    # (a bound against a pathological slow-down only; early release with ring traffic beside a
    # 2-layer step has measured 1.41x and 1.57x on different boxes)
    assert busy_ms < plain_ms * 2.0
