#!/usr/bin/env python
"""Headline benchmark: audio-seconds/s of CTC acoustic-model *training* on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one full training pass over one synthetic minibatch of raw 16 kHz int16 PCM that is
already resident in HBM: log-mel features + normalisation -> forward (conv front-end, BiLSTM
stack, dense, logits) -> fused log-softmax + CTC loss/gradient -> backward -> RCCL gradient
all-reduce (N > 1) -> TensorFlow-form Adam.

Workload (``--workload``, default ``c3``): BASELINE.json configs[2] - DS2, 2 conv layers +
5 x BiLSTM-1024, batch 32 per GPU, 10 s utterances (999 feature frames -> T' = 500), fp32.  It is
the per-GPU unit of configs[3] (global batch 256 on 8 GPUs), so the N = 1/2/4/8 curve is weak
scaling of exactly this workload.  ``--workload c2`` is configs[1] (2 x BiLSTM-1024, batch 16);
at N = 1 the default run measures it too and reports it under ``other_workloads``.

One process per GPU.  Started without a torchrun environment and with ``--gpus N`` > 1, this
script re-executes itself under ``torch.distributed.run`` with N ranks (backend nccl = RCCL) and
exits non-zero if fewer than N GPUs are visible - it never silently falls back to one GPU.

Rank 0 prints ONE JSON line (schema in the task contract) that also carries
  "roofline":     the dominant kernel (the recurrent time-step kernel) against its roof, from
                  HIP events recorded by the library around every launch in the timed region;
                  ``traffic`` comes from the committed rocprofv3 PMC summary
                  (profiles/pmc_traffic.json, written by tools/make_profile_md.py), and
  "cpu_baseline": the same graph in stock torch CPU operators (oracle/torch_ref.py, kind
                  "port" - the reference's TensorFlow cannot run offline) on the host cores,
                  rank 0, N = 1 only, SAME batch as the GPU line, bounded sample.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (conv_filters, rnn layers, hidden, dense, per-GPU batch, seconds, rnn_cell)
    'c2': ((32, 32), 2, 1024, 2048, 16, 10.0, 'lstm'),           # BASELINE.json configs[1]
    'c2_3conv': ((32, 32, 96), 2, 1024, 2048, 16, 10.0, 'lstm'),  # with the reference's 3 convs
    'c3': ((32, 32), 5, 1024, 2048, 32, 10.0, 'lstm'),            # configs[2] = per-GPU unit of [3]
    'c3_3conv': ((32, 32, 96), 5, 1024, 2048, 32, 10.0, 'lstm'),
    # the reference's own flag defaults (asr/params.py): 3 convs, 4 x ReLU-RNN-2048, batch 16
    'ref_default': ((32, 32, 96), 4, 2048, 2048, 16, 10.0, 'rnn_relu'),
    # the reference's best published model (testruns.md: 3c4r2d / 3c5r2d, LSTM cells, 2048 units)
    'ref_best': ((32, 32, 96), 4, 2048, 2048, 16, 10.0, 'lstm'),
    'tiny': ((8, 8), 1, 128, 128, 4, 2.0, 'lstm'),                # plumbing check
    # configs[4]: mixed-length bucketed batches (0.7-17 s, LibriSpeech-shaped), beam width 64;
    # `seconds` is a placeholder - every batch of the seeded bucket sequence has its own length
    'c5': ((32, 32), 2, 1024, 2048, 16, None, 'lstm'),
}
BASELINE_NAMES = {'c2': 'BASELINE.json configs[1]', 'c3': 'BASELINE.json configs[2]',
                  'c5': 'BASELINE.json configs[4]'}
C5_BEAM_WIDTH = 64

RELEASE_TEXT = {
    'held': 'after the last persistent recurrence launch of the step (hold_until=rnn0), in '
            'buckets of at most max_bucket_bytes',
    'early': 'per layer, behind its weight-gradient GEMMs on the side stream, beside the '
             'recurrences of the layers below (CTCASR_ALLREDUCE_EARLY=1, the Trainer\'s default)'}

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0
PMC_TRAFFIC_JSON = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')


def dtype_label(model):
    """The arithmetic the step computes in, spelled out: tensors, accumulation and the own kernels
    are fp32; the big GEMMs multiply fp16 / bf16 PIECES of the fp32 operands on the 16-bit matrix
    pipe unless CTCASR_SPLIT_GEMM=0."""
    which = [name for name, on in (('forward', model.rnn_fwd_f16), ('backward', model.rnn_bwd_f16))
             if on]
    rec = '; {} recurrence (h x W_hh / dgates x W_hh) as fp16x3 split, fp32 accumulate'.format(
        ' and '.join(which)) if which else ''
    if getattr(model, 'conv_f16', False) and model.cfg.used_model == 'ds2':
        rec += '; the convolutions (forward, data gradient, kernel gradient) as fp16x3 split'
    if not model.split_gemm:
        return 'f32' if not rec else 'f32 tensors / accumulate / GEMMs' + rec
    gemms = 'bf16x6 split (24 significand bits)'
    if model.fwd_f16:
        gemms = 'fp16x3 split (22 significand bits) where the layer input is bounded{} - {} - ' \
                'bf16x6 split (24 bits) elsewhere'.format(
                    ' or, behind a ReLU cell, scaled per row / per column on the device'
                    if getattr(model, 'unbounded_f16', False) else '',
                    'forward projections and their gradient GEMMs' if model.bwd_f16
                    else 'forward projections')
    return 'f32 tensors / accumulate / own kernels; projection GEMMs: ' + gemms + rec


def conv_flops_per_utt(cfg, frames):
    """Forward FLOPs of the convolution front end (SURVEY.md 8d formula's first term)."""
    from ctc_asr_amd.model import CONV_KERNEL_SIZES
    t_out = cfg.output_time(frames)
    flops, c_in, freq = 0.0, 1, cfg.num_features
    if cfg.used_model == 'ds2':
        for i, c_out in enumerate(cfg.conv_filters):
            freq = -(-freq // 2)
            k_t, k_f = CONV_KERNEL_SIZES[i]
            flops += 2.0 * t_out * freq * c_out * k_t * k_f * c_in
            c_in = c_out
    return flops


def forward_flops_per_utt(cfg, frames):
    """SURVEY.md 8d formula: conv + input/recurrent projections + dense4 + logits."""
    from ctc_asr_amd.model import GATES
    t_out = cfg.output_time(frames)
    flops = conv_flops_per_utt(cfg, frames)
    gates, hidden = GATES[cfg.cell], cfg.num_units_rnn
    in_size = cfg.rnn_input_size()
    for _ in range(cfg.num_layers_rnn):
        flops += 2.0 * t_out * 2 * gates * hidden * (in_size + hidden)
        in_size = 2 * hidden
    flops += 2.0 * t_out * 2 * hidden * cfg.num_units_dense
    flops += 2.0 * t_out * cfg.num_units_dense * cfg.num_classes
    return flops


def mixed_roof(cfg, frames, utterances, ms_per_step, split_gemm, world=1, fwd_f16=True,
               bwd_f16=True, rec_f16=(False, False), conv_f16=False, unbounded_f16=False):
    """Time one step would take with every FLOP at the peak of the pipe it runs on: the fp32
    matrix pipe (157.3 TF) for the own kernels; for the split GEMMs the 16-bit matrix pipe
    (2500 TF) divided by the products per fp32 product - three fp16 products where the layer's
    input is bounded (forward projections; with `bwd_f16` their gradient GEMMs too, dxw scaled per
    column / row), six bf16 products for the rest (dense4's kernel gradient; everything behind a
    ReLU-RNN cell)."""
    split, fp32 = training_flops_by_pipe(cfg, frames)
    if not split_gemm:
        split, fp32 = 0.0, split + fp32
    from ctc_asr_amd.model import GATES
    t_out = cfg.output_time(frames)
    rec = 2.0 * t_out * 2 * GATES[cfg.cell] * cfg.num_units_rnn ** 2 * cfg.num_layers_rnn
    dense4 = 2.0 * t_out * 2 * cfg.num_units_rnn * cfg.num_units_dense
    three = 0.0
    if split_gemm and fwd_f16 and (cfg.cell != 'rnn_relu' or unbounded_f16):
        three = (split - rec) / 3.0                       # the forward products
        if bwd_f16:
            three = split - dense4            # + every gradient GEMM but dense4's kernel gradient
    # the recurrences' own products (h W_hh^T forward, dgates W_hh backward: `rec` each) leave the
    # fp32 pipe for three fp16 products where the fp16-pipe persistent kernels run them
    # (LSTM: both passes have fp16-pipe kernels; GRU: the forward pass)
    rec16 = rec * sum(1 for on in rec_f16 if on) if cfg.cell in ('lstm', 'gru') else 0.0
    fp32 -= rec16
    # ... and the convolutions (forward, data gradient, kernel gradient: csrc/conv16.hip)
    conv16 = 3.0 * conv_flops_per_utt(cfg, frames) if (conv_f16 and cfg.used_model == 'ds2') \
        else 0.0
    fp32 -= conv16
    rec16 += conv16
    roof_ms = utterances / world * (
        (three + rec16) * 3.0 / (BF16_MFMA_PEAK_TFLOPS * 1e12) +
        (split - three) * 6.0 / (BF16_MFMA_PEAK_TFLOPS * 1e12) +
        fp32 / (FP32_MFMA_PEAK_TFLOPS * 1e12)) * 1e3
    return {'ms_per_step_at_peak': round(roof_ms, 3), 'frac': round(roof_ms / ms_per_step, 4),
            'fp32_equiv_tflop_split_gemms': round(utterances / world * split / 1e12, 3),
            'of_which_fp16x3': round(utterances / world * three / 1e12, 3),
            'tflop_recurrences_fp16x3': round(utterances / world * (rec16 - conv16) / 1e12, 3),
            'tflop_convolutions_fp16x3': round(utterances / world * conv16 / 1e12, 3),
            'tflop_fp32_pipe': round(utterances / world * fp32 / 1e12, 3),
            'note': 'per GPU; split GEMMs priced at 2500 TF / 3 (fp16 pieces: bounded layer inputs, '
                    'their gradient GEMMs with per-column / per-row scales of dxw; the recurrences\' '
                    'own products and the convolutions where the fp16-pipe kernels run) and '
                    '2500 TF / 6 (bf16 pieces), the other own kernels at 157.3 TF'}


BF16_MFMA_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA peak, MI355X_MICROARCH.md chip table


def training_flops_by_pipe(cfg, frames):
    """(fp32-equivalent FLOPs of one utterance's training step that run as bf16-split GEMMs -
    six bf16 products per fp32 product, ctc_asr_amd/split_gemm.py -, FLOPs that stay on the fp32
    matrix pipe inside the own kernels).  Split: the input projections and dense4 (forward, data
    and weight gradient) and the recurrent weight gradient; fp32: the recurrent products inside the
    persistent kernels (forward and data gradient), the convolutions, the logits layer."""
    from ctc_asr_amd.model import GATES
    t_out = cfg.output_time(frames)
    gates, hidden = GATES[cfg.cell], cfg.num_units_rnn
    proj, rec, in_size = 0.0, 0.0, cfg.rnn_input_size()
    for _ in range(cfg.num_layers_rnn):
        proj += 2.0 * t_out * 2 * gates * hidden * in_size
        rec += 2.0 * t_out * 2 * gates * hidden * hidden
        in_size = 2 * hidden
    dense4 = 2.0 * t_out * 2 * hidden * cfg.num_units_dense
    total = 3.0 * forward_flops_per_utt(cfg, frames)
    split = 3.0 * (proj + dense4) + rec
    return split, total - split


# ------------------------------------------------------------------------------ CPU baseline
def _cpu_baseline_worker(spec):
    """Runs in a child process: time the torch-CPU restatement (fwd + bwd + TF-Adam).  First a
    thread-count sweep on a small proxy batch, then the full batch with the best count."""
    from ctc_asr_amd.model import ModelConfig, init_params, to_oracle_layout
    from ctc_asr_amd.synth import synthetic_batch
    from oracle import torch_ref
    cfg = ModelConfig(**spec['cfg'])
    model = torch_ref.TorchRefModel(to_oracle_layout(init_params(cfg, 0), cfg), cfg.used_model,
                                    cfg.rnn_cell, cfg.cudnn)
    opt = torch_ref.TFAdam(model.parameters())

    def make_step(batch):
        feats, lengths, labels, _ = synthetic_batch(batch, spec['seconds'], seed=99,
                                                    frames=spec['frames'])
        label_rows = [[int(v) for v in row if v] for row in labels]
        feats_t = torch.tensor(feats)

        def one_step():
            opt.zero_grad()
            logits, seq_len = model(feats_t, lengths)
            loss, _ = model.loss(logits, seq_len, label_rows)
            loss.backward()
            opt.step()
        return one_step

    # the thread count is chosen AT THE MEASURED BATCH (a batch-4 proxy picked 16
    # threads for batch 32): one warm-up + one timed step per candidate, most promising first,
    # until the sweep budget is spent or a candidate is clearly slower than the best so far
    sweep = {}
    full = make_step(spec['batch'])
    t_start = time.perf_counter()
    warm = None
    for threads in spec['candidates']:
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        full()                                    # warm-up at this thread count
        warm = warm or time.perf_counter() - t0
        t0 = time.perf_counter()
        full()
        sweep[threads] = time.perf_counter() - t0
        spent = time.perf_counter() - t_start
        if spent + 2.2 * sweep[threads] > spec['sweep_budget_s'] or \
                sweep[threads] > 1.5 * min(sweep.values()):
            break
    best = min(sweep, key=sweep.get)
    torch.set_num_threads(best)
    steps = int(max(0, min(4, (spec['budget_s'] - (time.perf_counter() - t_start)) //
                           max(sweep[best], 1e-3))))
    total, count = sweep[best], 1
    for _ in range(steps):
        t0 = time.perf_counter()
        full()
        total += time.perf_counter() - t0
        count += 1
    print(json.dumps({'steps': count, 'per_step': total / count, 'warm': warm, 'threads': best,
                      'sweep': {str(k): round(v, 3) for k, v in sweep.items()}}))


T_PROCESS_START = time.time()


def host_cores():
    """(usable logical CPUs, physical cores among them)."""
    try:
        logical = len(os.sched_getaffinity(0))
    except AttributeError:
        logical = os.cpu_count() or 1
    physical = logical
    try:
        import psutil
        physical = min(logical, psutil.cpu_count(logical=False) or logical)
    except Exception:
        pass
    return logical, physical


def cpu_baseline(cfg_kwargs, batch, seconds, frames, budget_s=90.0, hard_limit_s=300.0):
    """CPU baseline (kind "port") at the SAME batch as the GPU line, in a child process with a
    hard time limit so that a slow host can never stall the benchmark.  The thread count is
    swept at that batch (torch's CPU LSTM stops scaling well before all cores of a big host):
    16, 32, 64, then 8 and everything, while the sweep budget lasts."""
    logical, physical = host_cores()
    candidates = [c for c in (16, 32, 64, 8, 128, physical) if c <= physical] or [1]
    candidates = list(dict.fromkeys(candidates))
    code = ('import json,sys; sys.path.insert(0, {!r}); import bench; '
            'bench._cpu_baseline_worker(json.loads(sys.argv[1]))').format(ROOT)
    spec = {'cfg': cfg_kwargs, 'candidates': candidates, 'batch': batch,
            'seconds': seconds, 'frames': frames,
            'budget_s': budget_s, 'sweep_budget_s': 75.0}
    try:
        out = subprocess.run([sys.executable, '-c', code, json.dumps(spec)],
                             capture_output=True, text=True, timeout=hard_limit_s)
        info = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception as err:      # timeout or failure: never block the GPU result
        return {'value': None, 'unit': 'audio-s/s', 'cores': physical, 'kind': 'port',
                'sample': 'CPU baseline did not finish within {:.0f} s ({})'.format(
                    hard_limit_s, type(err).__name__)}
    return {'value': round(batch * seconds / info['per_step'], 3), 'unit': 'audio-s/s',
            'cores': info['threads'], 'kind': 'port',
            'sample': '{} timed fwd+bwd+Adam step(s) after 1 warm-up, batch {} x {:.0f} s (the '
                      'GPU line\'s batch), torch {} CPU ops (oracle/torch_ref.py), {:.2f} s/step '
                      'with {} threads = fastest of a sweep over {} threads at this batch '
                      '(s/step {}; one warm-up step per thread count); host: {} physical / {} '
                      'logical cores'.format(
                          info['steps'], batch, seconds, torch.__version__, info['per_step'],
                          info['threads'], candidates, info['sweep'], physical, logical)}


# ------------------------------------------------------------------------------ measurement
def pmc_traffic(workload, which, launch_steps):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/pmc_traffic.json, written by tools/make_profile_md.py).  Returns a dict for the
    roofline object; ``traffic`` is None (with the reason) when no matching entry exists."""
    try:
        with open(PMC_TRAFFIC_JSON) as handle:
            table = json.load(handle)
    except (OSError, ValueError):
        return {'traffic': None, 'traffic_note': 'no profiles/pmc_traffic.json'}
    for entry in table.get('entries', []):
        # (a bucket sequence: the profiled pass and this run may average over different batches -
        # the bytes of a launch are proportional to its time steps)
        ratio = launch_steps / float(entry['steps_per_launch']) if workload == 'c5' else 1.0
        if entry['workload'] == workload and entry['pass'] == which and \
                (abs(entry['steps_per_launch'] - launch_steps) <= 1 or
                 (workload == 'c5' and 0.8 < ratio < 1.25)):
            fetch, write = entry['fetch_kb'] * 1024.0 * ratio, entry['write_kb'] * 1024.0 * ratio
            out = {
                # FETCH_SIZE reads half of a wide coalesced stream on gfx950 (MI355X_MICROARCH.md
                # section HBM): the corrected figure doubles the read side
                'traffic': int(2 * fetch + write),
                'traffic_raw': int(fetch + write),
                'traffic_note': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of '
                                'this workload, per launch; traffic = 2 x FETCH + WRITE (gfx950 '
                                'FETCH_SIZE correction), traffic_raw = as reported; {}'.format(
                                    entry.get('source', ''))}
            if entry.get('algorithmic_bytes'):
                out['algorithmic_bytes_per_launch'] = int(entry['algorithmic_bytes'] * ratio)
                out['traffic_over_algorithmic'] = round(out['traffic'] /
                                                        (entry['algorithmic_bytes'] * ratio), 2)
            return out
    return {'traffic': None,
            'traffic_note': 'no PMC entry for ({}, {}, {} steps per launch) in '
                            'profiles/pmc_traffic.json'.format(workload, which, launch_steps)}


def measure(name, args, rank, local_rank, world, allreduce_early=None, reduce=True,
            model_attrs=None, stand_in=None):
    """Build the workload ``name`` on this rank's GPU, run warm-up + timed steps under the
    contract's protocol and return the result dict (rank 0) or None.  ``allreduce_early``
    selects the release mode of the gradient buckets (None: the environment / default),
    ``reduce=False`` runs the same step without any collective (to price the all-reduce),
    ``model_attrs`` overrides attributes of the CTCModel (which arithmetic the GEMMs run in:
    PROBE_VARIANTS)."""
    from ctc_asr_amd import hip
    from ctc_asr_amd.engine import Trainer
    from ctc_asr_amd.model import CTCModel, GATES, ModelConfig
    from ctc_asr_amd.synth import random_pcm, synthetic_batch
    import torch.distributed as dist

    device = 'cuda:{}'.format(local_rank)
    filters, layers, hidden, dense, batch, seconds, rnn_cell = WORKLOADS[name]
    cfg = ModelConfig(used_model='ds2', conv_filters=filters, num_units_dense=dense,
                      num_layers_rnn=layers, num_units_rnn=hidden, rnn_cell=rnn_cell, cudnn=True,
                      dense_dropout_rate=args.dropout)
    # fixed input shape: let MIOpen benchmark its convolution kernels once (warm-up steps)
    trainer = Trainer(cfg, device=device, seed=0, world_size=world, rank=rank, conv_autotune=True,
                      allreduce_early=allreduce_early, reduce=reduce,
                      collective_stand_in=stand_in)
    model = trainer.model
    for key, value in (model_attrs or {}).items():    # (None: the defaults / CTCASR_* switches)
        setattr(model, key, value)
    if args.rnn_bwd_whole_chip:
        model.rnn_bwd_flags = hip.RNN_WHOLE_CHIP

    # synthetic 16 kHz utterances: int16 PCM resident in HBM (SURVEY.md 8d recipe), random labels
    _, _, labels, _ = synthetic_batch(batch, seconds, seed=1234 + rank, frames=1)
    rng = np.random.default_rng(4321 + rank)
    num_samples = int(round(seconds * 16000))
    pcm_d = torch.from_numpy(np.stack([random_pcm(rng, num_samples) for _ in range(batch)])) \
        .to(device)
    nsamp_d = torch.full((batch,), num_samples, dtype=torch.int32, device=device)
    frames = hip.features_num_frames(num_samples)
    feat_buf = torch.empty((batch, frames, 80), dtype=torch.float32, device=device)
    len_d = torch.empty(batch, dtype=torch.int32, device=device)
    packed = CTCModel.pack_labels(labels, model.device)

    def step():
        # hot path from raw audio: log-mel features + per-utterance normalisation on the GPU,
        # then forward / CTC / backward / all-reduce / Adam
        hip.features(pcm_d, nsamp_d, 'mel', 'local', False, 16000, out=feat_buf, out_len=len_d)
        # (check=True: the production path of train.py - its error checks are deferred and do
        # not stall the host, engine.Trainer.train_step)
        # (at N > 1 nothing may raise on one rank only in the middle of the loop: the checks run
        # at the two synchronisation points below and their outcome is agreed on by all ranks)
        return trainer.train_step(feat_buf, len_d, packed,
                                  check=not args.no_step_checks and world == 1)

    failure = []

    def checks():
        """Errors of the steps so far; at N > 1 a failed leg (e.g. a recurrence time-out under
        the early release mode) is reported in the line instead of killing the other legs."""
        try:
            trainer.drain_checks()
            CTCModel.check_status(model.last_status)
            # a persistent recurrence launch that gave up at a grid barrier produced garbage
            # (sticky word: covers every layer, pass and step since the previous check)
            model.check_rnn_error()
        except (hip.CtcAsrError, ValueError) as err:
            if world == 1:
                raise
            failure.append('rank {}: {}'.format(rank, err))

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    checks()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    hip.EVENTS = {}
    hip.set_option('rnn_kernel_events', 1)      # event pairs right around the persistent kernels
    hip.rnn_kernel_events()
    trainer.reducer.launched = 0
    trainer.host_wait_s = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    # host side done enqueuing; minus the time it spent waiting for the GPU to come within
    # Trainer.max_steps_ahead steps = what the host needs to enqueue the steps
    issued = time.perf_counter() - t0 - trainer.host_wait_s
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    events = hip.drain_events()
    hip.EVENTS = None
    kernel_events = hip.rnn_kernel_events()
    hip.set_option('rnn_kernel_events', 0)
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=device)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        rank_ms = [float(t.item()) / args.steps * 1e3 for t in every]
        elapsed = max(float(t.item()) for t in every)          # MAX over ranks (the contract)
    checks()
    # (with --no-step-checks / at N > 1 the per-step checks are off: a sticky guard word would make
    # every later step a no-op that is still timed - never report such a run as throughput)
    skipped = trainer.skipped_step_count()
    if skipped:
        failure.append('rank {}: {} training step(s) were dropped on the device (guard word set)'
                       .format(rank, skipped))
        if world == 1:
            raise hip.CtcAsrError(failure[-1])
    failed_ranks = 0
    if world > 1:
        flag = torch.tensor([1.0 if failure else 0.0], device=device)
        dist.all_reduce(flag)
        failed_ranks = int(flag.item())

    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        audio_s = world * batch * seconds
        value = audio_s / (elapsed / args.steps)
        t_out = cfg.output_time(frames)
        flops_per_audio_s = 3.0 * forward_flops_per_utt(cfg, frames) / seconds
        gates = GATES[cfg.cell]
        # dominant kernel: the recurrence.  Persistent variant: one launch runs a range of steps
        # with the recurrent weights resident in LDS -> fp32-MFMA roof (the limiter is the
        # per-step all-to-all exchange of h / dgates).  Streaming variant: one launch per time
        # step re-reads the weights -> HBM/MALL bandwidth roof.
        persistent = hip.rnn_persistent_supported(cfg.cell, t_out, batch, hidden)
        roofline = None
        dom = max(('rnn_fwd', 'rnn_bwd'), key=lambda k: events.get(k, (0, 0.0))[1])
        calls, dom_ms = events.get(dom, (0, 0.0))
        if calls:
            flops_per_step = 2.0 * 2 * batch * hidden * gates * hidden    # both directions
            bytes_per_step = 2 * (gates * hidden * hidden + batch * hidden * (2 + gates)) * 4
            if persistent:
                # the kernel alone (library-side event pair), not the call incl. its memsets
                calls, dom_ms = kernel_events[dom[4:]]
                avg_s = dom_ms * 1e-3 / calls
                # a layer's recurrence may be cut into several launches (CTCModel.bwd_chunks /
                # fwd_chunks); a launch then covers T' / chunks time steps
                launch_steps = t_out * args.steps * cfg.num_layers_rnn / float(calls)
                achieved = flops_per_step * launch_steps / avg_s / 1e12
                # which matrix pipe the kernel's product runs on: fp32 MFMA, or - the fp16-pipe
                # kernels - three fp16 products per fp32 product (peak 2500 / 3 fp32-equivalent)
                f16_kernel = model.arithmetic().get(
                    'rnn0/recurrence_{}'.format(dom[4:])) == 'fp16x3'
                peak = BF16_MFMA_PEAK_TFLOPS / 3.0 if f16_kernel else FP32_MFMA_PEAK_TFLOPS
                # 17..32 rows, LSTM-1024, cuDNN semantics, half of the chip: the staggered-tile
                # kernel (prnn_bwd16s_kernel) unless CTCASR_RNN_STAGGER=0
                staggered = (dom == 'rnn_bwd' and f16_kernel and rnn_cell == 'lstm' and
                             hidden == 1024 and 16 < batch <= 32 and batch % 8 == 0 and cfg.cudnn and
                             not args.rnn_bwd_whole_chip and
                             bool(getattr(model, 'rnn_stagger_flag', 0)))
                # the kernel's name as a trace shows it: H = 2048 LSTM backward = prnn_bwd16w_kernel
                # (one direction per launch; K split over pairs of workgroups unless
                # CTCASR_RNN_KPAIR_2048=0), the ReLU cell's fp16 backward = prnn_relu16_kernel<true>
                kname = 'prnn_{}{}_kernel<{}>'.format(
                    dom[4:], ('16s' if staggered else '16') if f16_kernel else '', rnn_cell.upper())
                if f16_kernel and dom == 'rnn_bwd' and rnn_cell == 'lstm' and hidden == 2048:
                    kname = 'prnn_bwd16w_kernel<LSTM-2048{}>'.format(
                        ', K split over pairs of workgroups'
                        if getattr(model, 'rnn_kpair_wide', False) else '')
                elif f16_kernel and rnn_cell == 'rnn_relu':
                    kname = 'prnn_relu16_kernel<{}>'.format('true' if dom == 'rnn_bwd' else 'false')
                roofline = {
                    'kernel': '{} (persistent, LDS-resident recurrent weights; '
                              'one launch = {:.0f} time steps x 2 directions, batch {}{})'.format(
                                  kname, launch_steps, batch,
                                  '; the two 16-row tiles staggered by half a step'
                                  if staggered else ''),
                    'bound': 'mfma', 'achieved': round(achieved, 2),
                    'peak': round(peak, 1), 'unit': 'TFLOP/s',
                    'frac': round(achieved / peak, 4),
                    'pipe': 'fp16 MFMA, 3 piece products per fp32 product (peak = 2500 / 3 '
                            'fp32-equivalent TFLOP/s)' if f16_kernel else 'fp32 MFMA',
                    'frac_of_fp32_mfma_peak': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4)}
                if dom == 'rnn_bwd':
                    # what actually bounds the all-gather backward recurrence: every workgroup
                    # pulls the dgates of ALL 4H gate columns of its rows through its CU's load
                    # path every step: bytes per CU and time step, and the rate
                    rows_per_cu = batch if not args.rnn_bwd_whole_chip else min(batch, 16)
                    per_cu = rows_per_cu * gates * hidden * 4.0
                    roofline['exchange_load_path'] = {
                        'bytes_per_cu_per_time_step': per_cu,
                        'gb_per_s_per_cu': round(per_cu * launch_steps / avg_s / 1e9, 1),
                        # a CU's vector-memory path moves 64 B per clock (MI355X_MICROARCH.md)
                        'peak_gb_per_s_per_cu': 153.6,
                        'frac': round(per_cu * launch_steps / avg_s / 1e9 / 153.6, 4),
                        'note': 'all-gather of dgates: B x 4H x 4 bytes per workgroup and step '
                                'whatever the decomposition; a CU sustains ~100 - 125 GB/s on '
                                'freshly written cross-XCD data (L1 path peak 64 B/clk = 150)'}
                roofline.update(pmc_traffic(name, dom, round(launch_steps)))
                # the H=1024 backward kernel runs on half the chip by default: the weight-gradient
                # GEMMs of the steps it has finished fill the other 128 CUs
                cus = 128 if (dom == 'rnn_bwd' and hidden == 1024 and
                              not args.rnn_bwd_whole_chip) else 256
                # (achieved against the peak of the kernel's OWN pipe on the CUs it holds - the
                # fp16 pipe's fp32-equivalent peak for the fp16x3 kernels)
                roofline.update({
                    'cus_occupied': cus,
                    'frac_of_pipe_peak_of_occupied_cus': round(
                        achieved / (peak * cus / 256.0), 4)})
                roofline.update({
                    'avg_launch_us': round(avg_s * 1e6, 1), 'launches': calls,
                    'algorithmic_flops_per_launch': flops_per_step * launch_steps,
                    'us_per_time_step': round(avg_s * 1e6 / launch_steps, 3),
                    'share_of_step': round(dom_ms / (elapsed * 1e3), 3),
                    'other_pass': {
                        'kernel': 'prnn_{}{}_kernel'.format(
                            'fwd' if dom == 'rnn_bwd' else 'bwd',
                            '16' if model.arithmetic().get('rnn0/recurrence_{}'.format(
                                'fwd' if dom == 'rnn_bwd' else 'bwd')) == 'fp16x3' else ''),
                        'us_per_time_step': round(
                            kernel_events['fwd' if dom == 'rnn_bwd' else 'bwd'][1] * 1e3 /
                            max(1.0, t_out * args.steps * cfg.num_layers_rnn), 3)}})
            else:
                launches = calls * t_out
                avg_s = dom_ms * 1e-3 / launches
                achieved_gbs = bytes_per_step / avg_s / 1e9
                roofline = {
                    'kernel': 'rnn_{}_step_kernel<{}> (one launch per time step)'.format(
                        dom[4:], rnn_cell.upper()),
                    'bound': 'hbm', 'achieved': round(achieved_gbs, 1), 'peak': HBM_PEAK_GBS,
                    'unit': 'GB/s', 'frac': round(achieved_gbs / HBM_PEAK_GBS, 4),
                    'traffic': None, 'avg_launch_us': round(avg_s * 1e6, 3),
                    'launches': launches, 'algorithmic_bytes_per_launch': bytes_per_step,
                    'mfma_tflops': round(flops_per_step / avg_s / 1e12, 2),
                    'share_of_step': round(dom_ms / (elapsed * 1e3), 3)}
        result = {
            'metric': 'audio-seconds/s training throughput (DS2, 10 s utterances)',
            'value': round(value, 2), 'unit': 'audio-s/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(ms_per_step, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': dtype_label(model), 'data': 'synthetic (16 kHz int16 Gaussian-noise PCM of fixed-length '
                                    'utterances resident in HBM, random labels at 15 chars/s)',
            'config': {'workload': '{}: DS2 {}-conv + {}xBi{}-{}, '
                                   'batch {}/GPU, {:.0f} s utterances'.format(
                                       BASELINE_NAMES.get(name, name), len(filters), layers,
                                       {'lstm': 'LSTM', 'rnn_relu': 'RNN(relu)'}[rnn_cell],
                                       hidden, batch, seconds),
                       'name': name, 'global_batch': world * batch, 'frames': frames,
                       'ctc_steps': t_out, 'parallelism': 'dp{}'.format(world),
                       'dense_dropout_rate': args.dropout,
                       'parameters': model.arena.num_parameters()},
            'loss': round(float(loss), 4),
            'step_tflops_fp32': round(value * flops_per_audio_s / 1e12, 2),
            'frac_of_fp32_mfma_peak': round(value * flops_per_audio_s / 1e12 /
                                            (FP32_MFMA_PEAK_TFLOPS * world), 4),
            'gemm_path': ('split: fp32 operands as three bf16 pieces each, the six products of '
                          'order <= 2 accumulated in fp32 on the 16-bit matrix pipe (gradient GEMMs; '
                          'closer to fp64 than the fp32 GEMM){}; profiles/r03_gemm_bf16_split.md; '
                          'CTCASR_SPLIT_GEMM=0 selects the fp32 library GEMMs'.format(
                              '; where the layer input is bounded (|h| <= 1, clipped ReLU): two '
                              'fp16 pieces and three products, the fp32 GEMM\'s error - forward '
                              'projections{}'.format(
                                  ' and their gradient GEMMs (dxw scaled per column / per row on '
                                  'the device)' if model.bwd_f16 else '') if model.fwd_f16 else ''))
                         if model.split_gemm else 'fp32 library GEMMs (CTCASR_SPLIT_GEMM=0)',
            'roof': mixed_roof(cfg, frames, batch * world, ms_per_step, model.split_gemm, world,
                               model.fwd_f16, model.bwd_f16,
                               (model.arithmetic().get('rnn0/recurrence_fwd') == 'fp16x3',
                                model.arithmetic().get('rnn0/recurrence_bwd') == 'fp16x3'),
                               conv_f16=model.arithmetic().get('conv0/forward') == 'fp16x3',
                               unbounded_f16=model.unbounded_f16),
            'kernel_ms_per_step': {k: round(v[1] / args.steps, 3) for k, v in events.items()},
            # time the host needed to enqueue a step; close to ms_per_step = launch-bound
            'host_enqueue_ms_per_step': round(issued / args.steps * 1e3, 3),
            'hbm_reserved_gb': round(torch.cuda.max_memory_reserved(device) / 2.0 ** 30, 2),
            'roofline': roofline,
        }
        if events.get('dgrad16', (0, 0.0))[0]:
            # the own block-scaled data-gradient kernel (csrc/dgrad16.hip), HIP event pairs around
            # its launches inside the timed region: dx = dxw W_ih of every recurrent layer
            calls_dg, ms_dg = events['dgrad16']
            in_sizes = [cfg.rnn_input_size()] + [2 * hidden] * (cfg.num_layers_rnn - 1)
            flop = sum(2.0 * t_out * batch * 2 * gates * hidden * n for n in in_sizes) * args.steps
            result['data_gradient_kernel'] = {
                'kernel': 'dgrad16_bs_kernel (dxw read as the fp16 pieces the backward recurrence '
                          'published; block-scaled, LDS-DMA operands, no library GEMM)',
                'launches': calls_dg, 'avg_launch_us': round(ms_dg * 1e3 / calls_dg, 1),
                'bound': 'mfma', 'unit': 'TFLOP/s',
                'achieved': round(flop / (ms_dg * 1e-3) / 1e12, 1),
                'peak': round(BF16_MFMA_PEAK_TFLOPS / 3.0, 1),
                'frac': round(flop / (ms_dg * 1e-3) / 1e12 / (BF16_MFMA_PEAK_TFLOPS / 3.0), 4),
                'pipe': 'fp16 MFMA, 3 piece products per fp32 product',
                'share_of_step': round(ms_dg / (elapsed * 1e3), 3)}
        if world > 1:
            result['allreduce'] = {
                'backend': dist.get_backend() + (' (RCCL)' if dist.get_backend() == 'nccl' else ''),
                'bytes_per_step': int(model.arena.size * 4),
                'launches_per_step': trainer.reducer.launched / float(args.steps),
                'bucket_bytes': int(trainer.reducer.bucket_elems * 4),
                'max_bucket_bytes': int(trainer.reducer.max_bucket_elems * 4),
                'mode': trainer.release if reduce else 'stubbed',
                'release': RELEASE_TEXT[trainer.release] if reduce else
                           'no collective at all (timing reference; replicas drift apart)',
                'rank_ms_per_step': {'min': round(min(rank_ms), 3), 'max': round(max(rank_ms), 3),
                                     'all': [round(v, 3) for v in rank_ms]}}
            if failed_ranks:
                result['allreduce']['failed'] = '{} rank(s) reported an error; {}'.format(
                    failed_ranks, '; '.join(failure) or 'see the other ranks\' stderr')
    del trainer, model
    torch.cuda.empty_cache()
    return result, (cfg, frames, batch, seconds)


# ------------------------------------------------------------------------------ C5
def c5_bucket_sequence(batch, count, seed=1234, pool=4096, num_buckets=96):
    """A fixed, seeded sequence of ``count`` bucketed batches: ``pool`` utterance durations from
    the LibriSpeech-shaped log-normal of SURVEY.md 8d within [0.7, 17] s (utterances outside
    are dropped like the reference's corpus filter does), bucket boundaries
    picked like ``get_bucket_boundaries`` does from a length-sorted CSV (``num_buckets`` = 96,
    the reference default), a seeded shuffle, and the package's own grouping code
    (``input_functions._group_batches``: a bucket emits a batch when it holds ``batch``
    utterances).  Returns a list of int arrays of sample counts, one per batch."""
    import bisect
    import random
    from ctc_asr_amd import input_functions as inp
    from ctc_asr_amd.synth import librispeech_like_durations
    rng = np.random.default_rng(seed)
    seconds = np.sort(librispeech_like_durations(rng, pool, drop=True))
    samples = np.round(seconds * 16000).astype(np.int64)
    frames = [1 + int(np.ceil((n - 400) / 160.0)) for n in samples]
    lengths = [int(float('{:.4f}'.format(n / 16000.0)) / 0.01) for n in samples]
    step = len(lengths) // num_buckets
    boundaries = sorted({lengths[i] for i in range(step, len(lengths), step)})
    order = list(range(pool))
    random.Random(seed).shuffle(order)
    stream = ((None, None, None, frames[i], int(samples[i])) for i in order)
    out = []
    for group in inp._group_batches(stream, True, boundaries, batch):
        if len(group) == batch:
            out.append(np.array([item[4] for item in group], dtype=np.int32))
        if len(out) == count:
            break
    del bisect
    return out


def measure_c5(args, rank, local_rank, world):
    """BASELINE.json configs[4] on this rank's GPU: training over a fixed seeded sequence of
    ``args.steps`` bucketed mixed-length batches (one untimed pass to warm the allocator over the
    sequence's shapes, one timed pass), then evaluation-style decoding of the same batches with
    the beam search at width 64 (logits of all batches decoded in grouped launches)."""
    from ctc_asr_amd import hip
    from ctc_asr_amd.engine import Trainer
    from ctc_asr_amd.labels import encode
    from ctc_asr_amd.model import CTCModel, GATES, ModelConfig
    from ctc_asr_amd.synth import random_label, random_pcm
    import torch.distributed as dist

    device = 'cuda:{}'.format(local_rank)
    filters, layers, hidden, dense, batch, _, rnn_cell = WORKLOADS['c5']
    cfg = ModelConfig(used_model='ds2', conv_filters=filters, num_units_dense=dense,
                      num_layers_rnn=layers, num_units_rnn=hidden, rnn_cell=rnn_cell, cudnn=True,
                      dense_dropout_rate=args.dropout, beam_width=C5_BEAM_WIDTH)
    trainer = Trainer(cfg, device=device, seed=0, world_size=world, rank=rank)
    model = trainer.model
    sequence = c5_bucket_sequence(batch, args.steps)
    rng = np.random.default_rng(4321 + rank)
    noise = torch.from_numpy(random_pcm(rng, 272000 + 4096 * batch)).to(device)
    batches = []
    for nsamp in sequence:
        width = int(nsamp.max())
        pcm = torch.zeros((batch, width), dtype=torch.int16, device=device)
        rows = []
        for b, n in enumerate(nsamp):
            start = int(rng.integers(0, 4096 * batch))
            pcm[b, :n] = noise[start:start + int(n)]
            text = random_label(rng, max(1, int(n / 16000.0 * 15.0)))
            rows.append(list(encode(text)))
        batches.append({'pcm': pcm, 'nsamp': torch.from_numpy(nsamp).to(device),
                        'labels': CTCModel.pack_labels(rows, model.device),
                        'seconds': float(nsamp.sum()) / 16000.0,
                        'padded_seconds': batch * width / 16000.0,
                        't_out': cfg.output_time(hip.features_num_frames(width))})

    def train_pass():
        loss = None
        for item in batches:
            feats, lengths = hip.features(item['pcm'], item['nsamp'], 'mel', 'local', False, 16000)
            loss = trainer.train_step(feats, lengths, item['labels'],
                                      check=not args.no_step_checks)
        return loss

    train_pass()
    torch.cuda.synchronize()
    trainer.drain_checks()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    hip.set_option('rnn_kernel_events', 1)
    hip.rnn_kernel_events()
    trainer.host_wait_s = 0.0
    t0 = time.perf_counter()
    loss = train_pass()
    issued = time.perf_counter() - t0 - trainer.host_wait_s
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kernel_events = hip.rnn_kernel_events()
    hip.set_option('rnn_kernel_events', 0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    trainer.drain_checks()
    # (the arithmetic of the TRAINING passes: `arithmetic()` reports the last forward pass, and the
    # decode passes below run in eval mode - no backward recurrence in theirs)
    train_arith = model.arithmetic()

    # evaluation-style decode: forward in eval mode, logits kept, grouped beam-search launches
    def decode_pass():
        pending, t_fwd, t_dec, decoded_utts = [], 0.0, 0.0, 0
        torch.cuda.synchronize()
        for item in batches:
            t1 = time.perf_counter()
            feats, lengths = hip.features(item['pcm'], item['nsamp'], 'mel', 'local', False, 16000)
            logits, seq_len = model.inference_fn(feats, lengths, training=False)
            pending.append((logits.clone(), seq_len.clone(), None))
            torch.cuda.synchronize()
            t_fwd += time.perf_counter() - t1
            group = model.decode_group_size(max(int(p[0].shape[0]) for p in pending), batch,
                                            beam_width=C5_BEAM_WIDTH)
            if len(pending) >= group or item is batches[-1]:
                t1 = time.perf_counter()
                results = model.decode_many(pending, beam_width=C5_BEAM_WIDTH)
                t_dec += time.perf_counter() - t1
                decoded_utts += sum(len(r[0]) for r in results)
                pending = []
        return t_fwd, t_dec, decoded_utts

    decode_pass()                       # warm-up (tree pools, shapes)
    t_fwd, t_dec, decoded_utts = decode_pass()
    model.check_rnn_error()

    result = None
    if rank == 0:
        audio_s = world * sum(item['seconds'] for item in batches)
        padded_s = world * sum(item['padded_seconds'] for item in batches)
        steps = len(batches)
        gates = GATES[cfg.cell]
        bwd_calls, bwd_ms = kernel_events['bwd']
        fwd_calls, fwd_ms = kernel_events['fwd']
        rnn_flops = sum(2.0 * 2 * batch * hidden * gates * hidden * item['t_out'] * layers
                        for item in batches)
        t_outs = [item['t_out'] for item in batches]
        arith = train_arith
        bwd_f16 = arith.get('rnn0/recurrence_bwd') == 'fp16x3'
        bwd_peak = BF16_MFMA_PEAK_TFLOPS / 3.0 if bwd_f16 else FP32_MFMA_PEAK_TFLOPS
        result = {
            'metric': 'audio-seconds/s training throughput (DS2, mixed-length bucketed batches '
                      '0.7-17 s)',
            'value': round(audio_s / elapsed, 2), 'unit': 'audio-s/s', 'n_gpus': world,
            'steps': steps, 'warmup': steps, 'ms_per_step': round(elapsed / steps * 1e3, 3),
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': dtype_label(model),
            'data': 'synthetic (16 kHz int16 Gaussian-noise PCM resident in HBM, durations from a '
                    'log-normal (median 10.5 s) filtered to [0.7, 17] s, 96 buckets, seeded order; random labels '
                    'at 15 chars/s)',
            'config': {'workload': '{}: DS2 {}-conv + {}xBiLSTM-{}, bucketed batches of {}/GPU, '
                                   'utterances 0.7-17 s, beam width {}'.format(
                                       BASELINE_NAMES['c5'], len(filters), layers, hidden, batch,
                                       C5_BEAM_WIDTH),
                       'name': 'c5', 'global_batch': world * batch,
                       'parallelism': 'dp{}'.format(world), 'dense_dropout_rate': args.dropout,
                       'bucket_sequence': {'batches': steps, 'seed': 1234, 'pool': 4096,
                                           'num_buckets': 96,
                                           'ctc_steps_min_mean_max': [min(t_outs), round(
                                               float(np.mean(t_outs)), 1), max(t_outs)],
                                           'audio_seconds': round(audio_s, 1),
                                           'padded_audio_seconds': round(padded_s, 1)},
                       'parameters': model.arena.num_parameters()},
            'loss': round(float(loss), 4),
            'padded_audio_s_per_s': round(padded_s / elapsed, 2),
            'host_enqueue_ms_per_step': round(issued / steps * 1e3, 3),
            'decode': {
                'beam_width': C5_BEAM_WIDTH, 'utterances': decoded_utts * world,
                'value': round(audio_s / t_dec, 2), 'unit': 'audio-s/s',
                'seconds_beam_search': round(t_dec, 4),
                'seconds_forward_incl_features': round(t_fwd, 4),
                'value_incl_forward': round(audio_s / (t_dec + t_fwd), 2),
                'note': 'evaluation path: eval-mode forward per batch, logits kept, beam search '
                        'in grouped launches (CTCModel.decode_many), host conversion included'},
            'roofline': {
                'kernel': 'prnn_bwd{}_kernel<LSTM> over the whole bucket sequence (launches of '
                          'T\'/2 steps x 2 directions, batch {})'.format(
                              '16' if bwd_f16 else '', batch),
                'bound': 'mfma', 'unit': 'TFLOP/s', 'peak': round(bwd_peak, 1),
                'pipe': 'fp16 MFMA, 3 piece products per fp32 product (peak = 2500 / 3 '
                        'fp32-equivalent TFLOP/s)' if bwd_f16 else 'fp32 MFMA',
                'achieved': round(rnn_flops / (bwd_ms * 1e-3) / 1e12, 2) if bwd_ms else None,
                'frac': round(rnn_flops / (bwd_ms * 1e-3) / 1e12 / bwd_peak, 4)
                if bwd_ms else None,
                'frac_of_fp32_mfma_peak':
                    round(rnn_flops / (bwd_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)
                    if bwd_ms else None,
                'launches': bwd_calls,
                'mean_time_steps_per_launch': round(sum(t_outs) * layers / max(bwd_calls, 1), 2),
                'avg_launch_us': round(bwd_ms * 1e3 / max(bwd_calls, 1), 1),
                'cus_occupied': 128,
                'other_pass': {'kernel': 'prnn_fwd{}_kernel'.format(
                                   '16' if arith.get('rnn0/recurrence_fwd') == 'fp16x3' else ''),
                               'launches': fwd_calls,
                               'achieved': round(rnn_flops / (fwd_ms * 1e-3) / 1e12, 2)
                               if fwd_ms else None}},
        }
    if result is not None and 'mean_time_steps_per_launch' in result.get('roofline', {}):
        # (PMC passes over this very bucket sequence: mean bytes per launch, mean steps per launch)
        result['roofline'].update(pmc_traffic('c5', 'rnn_bwd',
                                              result['roofline']['mean_time_steps_per_launch']))
    del trainer, model
    torch.cuda.empty_cache()
    return result


def measure_c5_from_disk(args, local_rank, utterances=416, repeat=8, log_frequency=200):
    """`train.py`'s own loop over a corpus ON DISK: C5-shaped utterances
    (0.7 - 17 s, LibriSpeech-like) written as WAV + CSV in the reference's layout to a temporary
    directory - ``utterances`` files, each listed ``repeat`` times in the manifest so that an
    epoch has about as many steps as the reference's logging period - driven through
    ``input_fn_generator('train_bucket')`` - CSV, WAV headers, shuffle buffer, buckets, reader
    pool, pinned uploads, features on the GPU - and `train.train_epoch` with the reference's
    logging cadence (``log_frequency`` = 200, asr/params.py:114: the first step and every 200th:
    loss read-back, beam-64 decode of the batch, edit distance / WER, summary records).  One
    untimed epoch (allocator, page cache), then the SAME batches once from HBM (raw PCM kept on the
    device, the step `measure_c5` times) and once more through the whole pipeline (same shuffle
    seed).  ``value`` = audio seconds of the epoch / its wall time; ``ratio_to_in_hbm`` is the
    number the item asks for (>= 0.9)."""
    import tempfile
    from ctc_asr_amd import hip, summaries, synth, train
    from ctc_asr_amd.engine import Trainer
    from ctc_asr_amd.input_functions import input_fn_generator
    from ctc_asr_amd.model import ModelConfig
    from ctc_asr_amd.params import CSV_DELIMITER, CSV_FIELDNAMES, FLAGS
    device = 'cuda:{}'.format(local_rank)
    filters, layers, hidden, dense, batch, _, rnn_cell = WORKLOADS['c5']
    with tempfile.TemporaryDirectory() as tmp:
        corpus, csv = os.path.join(tmp, 'corpus'), os.path.join(tmp, 'train.csv')
        rng = np.random.default_rng(77)
        rows = synth.write_corpus(corpus, csv,
                                  synth.librispeech_like_durations(rng, utterances, drop=True),
                                  seed=78, subdir='train', sacrificial_row=False)
        with open(csv, 'w', encoding='utf-8') as handle:      # every file `repeat` times, by length
            handle.write(CSV_DELIMITER.join(CSV_FIELDNAMES) + '\n')
            for row in [r for r in rows for _ in range(repeat)] + [rows[-1]]:
                handle.write(CSV_DELIMITER.join(row) + '\n')
        FLAGS.reset()
        FLAGS.update(corpus_dir=corpus, train_csv=csv, train_dir=os.path.join(tmp, 'ckpt'),
                     batch_size=batch, num_buckets=8, feature_type='mel',
                     feature_normalization='local', beam_width=C5_BEAM_WIDTH,
                     log_frequency=log_frequency, random_seed=5)
        cfg = ModelConfig(used_model='ds2', conv_filters=filters, num_units_dense=dense,
                          num_layers_rnn=layers, num_units_rnn=hidden, rnn_cell=rnn_cell,
                          cudnn=True, dense_dropout_rate=args.dropout, beam_width=C5_BEAM_WIDTH)
        trainer = Trainer(cfg, device=device, seed=0)
        kept = []
        for item in input_fn_generator('train_bucket', device=device, seed=11)():      # epoch A
            trainer.train_step(item.features['spectrogram'], item.features['spectrogram_length'],
                               item.packed_labels)
            kept.append((item.pcm, item.num_samples, item.packed_labels, item.audio_seconds))
        torch.cuda.synchronize()
        trainer.drain_checks()

        def replay():
            for pcm, nsamp, labels, _ in kept:
                feats, lengths = hip.features(pcm, nsamp, 'mel', 'local', False, 16000)
                trainer.train_step(feats, lengths, labels)
        replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        replay()
        torch.cuda.synchronize()
        in_hbm_s = time.perf_counter() - t0
        trainer.drain_checks()
        writer = summaries.SummaryWriter(FLAGS.train_dir, 'train')
        t0 = time.perf_counter()
        steps, _ = train.train_epoch(trainer, 'train_bucket', 2, 0, 1, writer, seed=11)
        torch.cuda.synchronize()
        disk_s = time.perf_counter() - t0
        FLAGS.reset()
    audio_s = sum(k[3] for k in kept)
    assert steps == len(kept)
    del trainer
    torch.cuda.empty_cache()
    return {'value': round(audio_s / disk_s, 2), 'unit': 'audio-s/s', 'steps': steps,
            'ms_per_step': round(disk_s / steps * 1e3, 3),
            'in_hbm_value': round(audio_s / in_hbm_s, 2),
            'ratio_to_in_hbm': round(in_hbm_s / disk_s, 4),
            'config': {'wav_files_on_disk': utterances, 'manifest_rows': utterances * repeat,
                       'batch': batch, 'num_buckets': 8,
                       'log_frequency': log_frequency, 'beam_width_of_logged_decodes': C5_BEAM_WIDTH,
                       'workload': 'BASELINE.json configs[4] on one GPU, corpus on disk: '
                                   'train.train_epoch over input_fn_generator(\'train_bucket\')'},
            'note': 'wall time of one epoch of train.py\'s loop incl. its logged steps (loss '
                    'read-back, beam-64 decode, metrics, summary records) against the same batches '
                    'stepped from raw PCM resident in HBM'}


# ------------------------------------------------------------------------------ parity probe
# small shape: 4 x 100 rows - enough for the model to take its split-GEMM path; the fall-back when
# the host cannot finish the full-size oracle in time
PROBE_BATCH, PROBE_FRAMES, PROBE_LABEL_LEN = 4, 199, 30        # T' = 100
# which GEMM arithmetic a probe variant runs: attributes of CTCModel
PROBE_VARIANTS = (
    ('default', {}),
    # (both: every own kernel on the fp32 pipe - recurrences and convolutions)
    ('bf16x6', {'fwd_f16': False, 'bwd_f16': False, 'rnn_fwd_f16': False, 'rnn_bwd_f16': False,
                'conv_f16': False}),
    ('fp32_library_gemms', {'split_gemm': False, 'rnn_fwd_f16': False, 'rnn_bwd_f16': False,
                            'conv_f16': False}),
)


def _probe_setup(cfg_kwargs, batch=PROBE_BATCH, frames=PROBE_FRAMES, label_len=PROBE_LABEL_LEN):
    """Model, parameters and inputs of the parity probe - deterministic, so that the GPU process
    and the CPU child build the same thing independently."""
    from ctc_asr_amd.model import ModelConfig, init_params
    cfg = ModelConfig(**cfg_kwargs)
    flat = init_params(cfg, 0)
    rng = np.random.default_rng(2024)
    for name in flat:       # the initialiser's biases are zero: give every term something to do
        flat[name] = (flat[name] + rng.normal(size=flat[name].shape) * 0.01).astype(np.float32)
    feats = rng.normal(size=(batch, frames, 80)).astype(np.float32)
    lengths = np.full(batch, frames, dtype=np.int32)
    labels = [[int(v) for v in rng.integers(1, 28, size=label_len)] for _ in range(batch)]
    return cfg, flat, feats, lengths, labels


def _parity_probe_worker(spec):
    """Child process: the oracle's side of the probe (float64 torch restatement on the CPU)."""
    from ctc_asr_amd.model import to_oracle_layout
    from oracle import torch_ref
    cfg, flat, feats, lengths, labels = _probe_setup(spec['cfg'], spec['batch'], spec['frames'],
                                                     spec['label_len'])
    torch.set_num_threads(spec['threads'])
    ref = torch_ref.TorchRefModel(to_oracle_layout(flat, cfg), cfg.used_model, cfg.rnn_cell,
                                  cfg.cudnn, dtype=torch.float64)
    t0 = time.perf_counter()
    with torch.no_grad():
        logits, seq_len = ref(torch.tensor(feats, dtype=torch.float64), lengths)
        loss, _ = ref.loss(logits, seq_len, labels)
    np.save(spec['logits_path'], logits.numpy())
    print(json.dumps({'loss': float(loss), 'seconds': time.perf_counter() - t0}))


def _probe_gpu_side(cfg_kwargs, device, batch, frames, label_len, variants):
    """Forward + CTC loss on the GPU for every GEMM-arithmetic variant: {name: (loss, logits,
    what ran)}."""
    from ctc_asr_amd.model import CTCModel
    cfg, flat, feats, lengths, labels = _probe_setup(cfg_kwargs, batch, frames, label_len)
    out = {}
    for name, attrs in variants:
        model = CTCModel(cfg, device, params=flat)
        for key, value in attrs.items():
            setattr(model, key, value)
        logits, seq_len = model.inference_fn(torch.tensor(feats), torch.tensor(lengths),
                                             training=True)
        model.loss_fn(logits, seq_len, labels)
        model.check_rnn_error()
        # (the mean of the fp32 per-utterance losses in float64: an fp32 mean of values near 1e3
        # resolves 1.2e-4 and would hide what the variants differ by)
        loss = float(model.last_per_utterance_loss.double().mean())
        out[name] = (loss, logits.cpu().numpy(), model.arithmetic())
        del model
        torch.cuda.empty_cache()
    return out


def parity_probe(cfg_kwargs, device, batch=PROBE_BATCH, frames=PROBE_FRAMES,
                 label_len=PROBE_LABEL_LEN, variants=PROBE_VARIANTS[:1], hard_limit_s=300.0):
    """The second half of BASELINE.json's metric ("CTC loss delta vs ref"): one forward + CTC
    loss of the workload's architecture on a fixed batch (seeded weights and inputs, dropout off)
    on the GPU - through the same kernels as the timed steps, once per GEMM-arithmetic variant -
    against the oracle (``oracle/torch_ref.py`` in float64, in a child process on the host that
    runs while the GPU does its side).  The oracle is the checker here, outside the timed
    region."""
    import tempfile
    _, physical = host_cores()
    with tempfile.TemporaryDirectory() as tmp:
        spec = {'cfg': cfg_kwargs, 'threads': min(32, physical), 'batch': batch, 'frames': frames,
                'label_len': label_len, 'logits_path': os.path.join(tmp, 'logits.npy')}
        code = ('import json,sys; sys.path.insert(0, {!r}); import bench; '
                'bench._parity_probe_worker(json.loads(sys.argv[1]))').format(ROOT)
        child = subprocess.Popen([sys.executable, '-c', code, json.dumps(spec)],
                                 stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        try:
            gpu = _probe_gpu_side(cfg_kwargs, device, batch, frames, label_len, variants)
            stdout, _ = child.communicate(timeout=hard_limit_s)
            answer = json.loads(stdout.strip().splitlines()[-1])
            ref_loss = answer['loss']
            ref_logits = np.load(spec['logits_path'])
        except Exception as err:        # noqa: BLE001 - report, never take the line down
            child.kill()
            child.communicate()
            return {'ctc_loss_delta': None, 'logits_max_abs_delta': None,
                    'batch': batch, 'frames': frames,
                    'note': 'oracle child failed: {}'.format(type(err).__name__)}
    per_variant = {}
    for name, (loss, logits, what) in gpu.items():
        per_variant[name] = {'ctc_loss_delta': abs(loss - ref_loss),
                             'ctc_loss_rel_delta': abs(loss - ref_loss) / max(1.0, abs(ref_loss)),
                             'logits_max_abs_delta': float(np.abs(logits - ref_logits).max()),
                             'loss_gpu': loss, 'arithmetic': what}
    first = per_variant[variants[0][0]]
    return {'ctc_loss_delta': first['ctc_loss_delta'],
            'logits_max_abs_delta': first['logits_max_abs_delta'],
            'loss_gpu': first['loss_gpu'], 'loss_oracle': ref_loss, 'tolerance': 1e-3,
            'batch': batch, 'frames': frames, 'ctc_steps': int(ref_logits.shape[0]),
            'labels_per_utterance': label_len, 'variants': per_variant,
            'oracle_seconds': round(answer.get('seconds', 0.0), 1), 'oracle_threads': spec['threads'],
            'note': 'forward + CTC loss of this workload\'s architecture at the shape named here, '
                    'seeded weights / inputs, dropout off, GPU vs oracle/torch_ref.py float64 on '
                    'the host (checker, outside the timed region); one GPU pass per GEMM '
                    'arithmetic (`variants`), the top-level deltas are the default path\'s'}


def merge_release_modes(legs, stub):
    """N > 1: fold the measurements of the release modes ({'held': line, 'early': line}) and of
    the stubbed step into ONE result line - the better mode's line, with every mode's step time,
    per-rank spread and exposed all-reduce time (= its step minus the stubbed step) under
    ``allreduce.modes``."""
    valid = [m for m in legs if 'failed' not in legs[m]['allreduce']] or list(legs)
    best = min(valid, key=lambda m: legs[m]['ms_per_step'])
    result = legs[best]
    result['allreduce']['modes'] = {
        m: {'ms_per_step': leg['ms_per_step'], 'value': leg['value'],
            'failed': leg['allreduce'].get('failed'),
            'mode': leg['allreduce']['mode'],     # early falls back to held at H = 2048
            'launches_per_step': leg['allreduce']['launches_per_step'],
            'rank_ms_per_step': leg['allreduce']['rank_ms_per_step'],
            'exposed_allreduce_ms': round(leg['ms_per_step'] - stub['ms_per_step'], 3),
            'exposed_allreduce_frac_of_step': round(
                (leg['ms_per_step'] - stub['ms_per_step']) / leg['ms_per_step'], 4)}
        for m, leg in legs.items()}
    result['allreduce']['chosen'] = best
    result['allreduce']['stubbed_ms_per_step'] = stub['ms_per_step']
    result['allreduce']['exposed_allreduce_ms'] = round(
        result['ms_per_step'] - stub['ms_per_step'], 3)
    result['allreduce']['exposed_allreduce_frac_of_step'] = round(
        (result['ms_per_step'] - stub['ms_per_step']) / result['ms_per_step'], 4)
    # (a mode in which any rank reported an error - e.g. a recurrence time-out under the early
    # release - is never the chosen one while another mode ran clean)
    result['allreduce']['choice_rule'] = 'fastest mode without a failed rank'
    # the number an N > 1 reader looks for first, right behind `value` in the printed line
    front = {}
    for key, val in result.items():
        front[key] = val
        if key == 'value':
            front['exposed_allreduce_frac_of_step'] = \
                result['allreduce']['exposed_allreduce_frac_of_step']
    return front


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(('127.0.0.1', 0))
        return sock.getsockname()[1]


def spawn_ranks(args):
    """``python bench.py --gpus N`` outside torchrun: start the N ranks (one per GPU, RCCL)."""
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if visible < args.gpus:
        sys.stderr.write('bench.py: --gpus {} requested but only {} GPU(s) visible; refusing to '
                         'run a smaller job under that label.\n'.format(args.gpus, visible))
        return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
        'HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
           '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
           '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='c3', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-workloads', action='store_true',
                    help='skip the secondary C2 measurement of the default N = 1 run')
    ap.add_argument('--leg-budget', type=float, default=900.0,
                    help='N > 1: seconds of wall clock after which no further release-mode leg is '
                         'started (the stubbed leg always runs)')
    ap.add_argument('--no-reference-models', action='store_true',
                    help='skip ref_default / ref_best (the reference\'s own configurations) among '
                         'the other workloads of the default N = 1 run')
    ap.add_argument('--dropout', type=float, default=0.1,
                    help='dense_dropout_rate (reference default 0.1)')
    ap.add_argument('--rnn-bwd-whole-chip', action='store_true',
                    help='persistent backward recurrence on all 256 CUs (default: 128)')
    ap.add_argument('--no-parity-probe', action='store_true',
                    help='skip the CTC-loss / logits delta against the oracle (N = 1)')
    ap.add_argument('--parity-probe', default='full', choices=('full', 'small'),
                    help='shape of the parity probe: the timed workload\'s own (default) or '
                         'batch 4 x T\' = 100')
    ap.add_argument('--parity-probe-limit', type=float, default=420.0,
                    help='seconds the float64 oracle child may take at full size')
    ap.add_argument('--no-step-checks', action='store_true',
                    help='train_step(check=False): without the deferred per-step error checks')
    ap.add_argument('--collective-stand-in', action='store_true',
                    help='N = 1 diagnostic: every gradient bucket launches a collective-shaped '
                         'kernel (24 resident workgroups holding their CUs for bytes / 150 GB/s on '
                         'a stream of its own, engine.GradientReducer(stand_in=)) - measures, for '
                         'both release modes, what ring kernels beside the persistent recurrences '
                         'cost; reported under `collective_stand_in`')
    ap.add_argument('--c5-batches', type=int, default=24,
                    help='bucketed batches in the C5 sequence of the default N = 1 run')
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))

    from ctc_asr_amd.engine import init_distributed
    import torch.distributed as dist

    world_env = int(os.environ.get('WORLD_SIZE', '1'))
    if world_env != args.gpus:
        raise SystemExit('--gpus {} does not match WORLD_SIZE {}'.format(args.gpus, world_env))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X; no GPU is visible.')
    # tools/two_rank_one_gpu.py (plumbing check of the N > 1 path on a one-GPU box: gloo, shared
    # device, streaming recurrence) sets this; the line then says so and means nothing else
    share_gpu = os.environ.get('CTCASR_BENCH_SHARE_GPU') == '1'
    if not share_gpu and \
            torch.cuda.device_count() < int(os.environ.get('LOCAL_WORLD_SIZE', world_env)):
        raise SystemExit('bench.py: {} ranks on this node but only {} GPU(s) visible.'.format(
            os.environ.get('LOCAL_WORLD_SIZE', world_env), torch.cuda.device_count()))
    rank, local_rank, world = init_distributed()
    torch.cuda.set_device(local_rank)

    ranks_seen, devices = 1, None
    if world > 1:
        # proof that the collective backend really connects `world` ranks, one GPU each
        ones = torch.ones(1, device='cuda:{}'.format(local_rank))
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        props = torch.cuda.get_device_properties(local_rank)
        mine = {'rank': rank, 'device': local_rank, 'name': props.name,
                'uuid': str(getattr(props, 'uuid', ''))}
        devices = [None] * world
        dist.all_gather_object(devices, mine)
        if ranks_seen != world or \
                (not share_gpu and len({(d['device'], d['uuid']) for d in devices}) != world):
            raise SystemExit('bench.py: ranks do not map to {} distinct GPUs: {}'.format(
                world, devices))

    def workload_cfg_kwargs(name):
        filters, layers, hidden, dense, _, _, rnn_cell = WORKLOADS[name]
        return dict(used_model='ds2', conv_filters=list(filters), num_units_dense=dense,
                    num_layers_rnn=layers, num_units_rnn=hidden, rnn_cell=rnn_cell,
                    cudnn=True, dense_dropout_rate=0.0)

    other, exit_code = {}, 0
    if args.workload == 'c5':
        result = measure_c5(args, rank, local_rank, world)
        cfg = frames = batch = seconds = None
        if world == 1:
            try:
                other['c5_from_disk'] = measure_c5_from_disk(args, local_rank)
            except Exception as err:        # noqa: BLE001
                other['c5_from_disk'] = {'error': '{}: {}'.format(type(err).__name__, err)}
    elif world == 1:
        result, (cfg, frames, batch, seconds) = measure(args.workload, args, rank, local_rank,
                                                        world)
    else:
        # N > 1: the driver runs this once per N, so the run itself is the experiment - both
        # release modes of the gradient buckets back to back (each: warm-up + K timed steps under
        # the contract's protocol), then the same step with the collectives stubbed out.  `value`
        # is the better mode's; everything measured is in `allreduce.modes`.
        forced = os.environ.get('CTCASR_ALLREDUCE_EARLY')
        legs, skipped = {}, []
        for mode in (('early',) if forced == '1' else ('held',) if forced == '0'
                     else ('held', 'early')):
            # (the run bounds itself: a leg that would start after --leg-budget seconds of wall
            # clock is skipped and named - every rank decides on rank 0's clock)
            late = torch.tensor([1.0 if legs and time.time() - T_PROCESS_START > args.leg_budget
                                 else 0.0], device='cuda:{}'.format(local_rank))
            dist.broadcast(late, src=0)
            if late.item():
                skipped.append(mode)
                continue
            legs[mode], shape = measure(args.workload, args, rank, local_rank, world,
                                        allreduce_early=(mode == 'early'))
        stub, _ = measure(args.workload, args, rank, local_rank, world, reduce=False)
        cfg, frames, batch, seconds = shape
        result = merge_release_modes(legs, stub) if rank == 0 else None
        if rank == 0 and skipped:
            result['allreduce']['skipped_modes'] = {
                m: 'not started: {:.0f} s of wall clock were spent (--leg-budget)'.format(
                    args.leg_budget) for m in skipped}
        if rank == 0 and 'failed' in result['allreduce']:
            exit_code = 4                # every measured mode failed: the line says why
    if world == 1 and args.workload == 'c3' and not args.no_other_workloads:
        # (secondary measurements: a failure here is reported in the line, it does not take the
        # headline measurement down with it)
        try:
            second, _ = measure('c2', args, rank, local_rank, world)
            other['c2'] = {k: second[k] for k in ('value', 'unit', 'ms_per_step', 'config',
                                                  'step_tflops_fp32', 'frac_of_fp32_mfma_peak',
                                                  'kernel_ms_per_step', 'roofline',
                                                  'host_enqueue_ms_per_step')}
        except Exception as err:        # noqa: BLE001 - anything: keep the headline line
            other['c2'] = {'error': '{}: {}'.format(type(err).__name__, err)}
        try:
            # the headline workload once more with the library's fp32 GEMMs (no bf16 split): what
            # the split buys, on this box, in this run
            for label, attrs in PROBE_VARIANTS[1:]:
                plain, _ = measure('c3', args, rank, local_rank, world, model_attrs=attrs)
                other['c3_' + label] = {k: plain[k] for k in (
                    'value', 'unit', 'ms_per_step', 'dtype', 'gemm_path', 'roof',
                    'kernel_ms_per_step', 'host_enqueue_ms_per_step')}
        except Exception as err:        # noqa: BLE001
            other['c3_variants'] = {'error': '{}: {}'.format(type(err).__name__, err)}
        try:
            c5_args = argparse.Namespace(**vars(args))
            c5_args.steps = args.c5_batches
            third = measure_c5(c5_args, rank, local_rank, world)
            other['c5'] = {k: third[k] for k in ('value', 'unit', 'ms_per_step', 'steps',
                                                 'config', 'padded_audio_s_per_s', 'decode',
                                                 'roofline', 'host_enqueue_ms_per_step')}
        except Exception as err:        # noqa: BLE001
            other['c5'] = {'error': '{}: {}'.format(type(err).__name__, err)}
        try:
            other['c5_from_disk'] = measure_c5_from_disk(args, local_rank)
        except Exception as err:        # noqa: BLE001
            other['c5_from_disk'] = {'error': '{}: {}'.format(type(err).__name__, err)}
    if world == 1 and args.workload == 'c3' and not args.no_other_workloads and \
            not args.no_reference_models:
        # the reference's own configurations: its flag defaults (asr/params.py:43-50: 3 conv + 4 x
        # ReLU-RNN-2048, batch 16) and its best published model (testruns.md: 3 conv + 4 x
        # BiLSTM-2048) - a few steps each, with the loss / logits deltas of that architecture
        # against the float64 oracle at the small probe shape (batch 4, T' = 100)
        ref_args = argparse.Namespace(**vars(args))
        ref_args.steps, ref_args.warmup = min(args.steps, 6), min(args.warmup, 2)
        for ref_name in ('ref_default', 'ref_best'):
            try:
                line, _ = measure(ref_name, ref_args, rank, local_rank, world)
                entry = {k: line[k] for k in ('value', 'unit', 'ms_per_step', 'steps', 'config',
                                              'dtype', 'kernel_ms_per_step', 'roofline')}
                if not args.no_parity_probe:
                    small = parity_probe(workload_cfg_kwargs(ref_name), 'cuda:{}'.format(local_rank))
                    entry['ctc_loss_delta'] = small['ctc_loss_delta']
                    entry['logits_max_abs_delta'] = small['logits_max_abs_delta']
                    entry['parity_probe'] = {k: small.get(k) for k in (
                        'loss_gpu', 'loss_oracle', 'batch', 'frames', 'ctc_steps', 'tolerance',
                        'note')}
                    entry['parity_probe']['arithmetic'] = \
                        (small.get('variants') or {}).get('default', {}).get('arithmetic')
                    if small['ctc_loss_delta'] is not None and (
                            small['ctc_loss_delta'] > 1e-3 * max(1.0, abs(small.get('loss_oracle')
                                                                       or 1.0)) or
                            small['logits_max_abs_delta'] > 1e-3):
                        exit_code = 3
                other[ref_name] = entry
            except Exception as err:        # noqa: BLE001
                other[ref_name] = {'error': '{}: {}'.format(type(err).__name__, err)}
    if world == 1 and args.collective_stand_in and args.workload != 'c5':
        # one GPU cannot run RCCL beside itself; its CU footprint can be stood in for
        report = {'stand_in': '24 workgroups x 256 threads per bucket, resident for bytes / '
                              '150 GB/s, own stream, waited for like an async all-reduce; '
                              '`*_traffic`: the same workgroups stream 6 x the bucket\'s bytes '
                              'through L2 / fabric / HBM meanwhile (a ring all-reduce\'s reads and '
                              'writes: ctcasr_collective_traffic)',
                  'plain_ms_per_step': result['ms_per_step']}
        for mode in ('held', 'early', 'held_traffic', 'early_traffic'):
            try:
                leg, _ = measure(args.workload, args, rank, local_rank, world,
                                 allreduce_early=mode.startswith('early'),
                                 stand_in=(24, 150.0, mode.endswith('traffic')))
                report[mode] = {'ms_per_step': leg['ms_per_step'],
                                'inflation_ms': round(leg['ms_per_step'] - result['ms_per_step'], 3),
                                'inflation_frac': round(leg['ms_per_step'] / result['ms_per_step']
                                                        - 1.0, 4),
                                'backward_recurrence_us_per_time_step':
                                    leg['roofline'].get('us_per_time_step')}
            except Exception as err:        # noqa: BLE001 - a time-out is the finding
                report[mode] = {'error': '{}: {}'.format(type(err).__name__, err)}
        result['collective_stand_in'] = report
    if rank == 0:
        if world > 1:
            result['allreduce']['ranks_seen_by_allreduce'] = ranks_seen
            result['devices'] = devices
            if share_gpu:
                result['devices_shared'] = True      # not a measurement: plumbing check only
        if other:
            result['other_workloads'] = other
        if world == 1 and not args.no_parity_probe:
            # BASELINE.json's metric, second half: CTC loss (and logits) delta vs the oracle, AT
            # THE SHAPE THE NUMBER ABOVE WAS MEASURED ON (C3: 5 layers, batch 32, 999 frames ->
            # T' = 500), once per GEMM arithmetic; the small shape only if the host cannot finish
            # the float64 oracle in time (or with --parity-probe small)
            probe_kwargs = workload_cfg_kwargs(args.workload)
            probe = None
            try:
                if args.parity_probe == 'full' and args.workload != 'c5':
                    probe = parity_probe(probe_kwargs, 'cuda:{}'.format(local_rank), batch=batch,
                                         frames=frames, label_len=int(round(15 * seconds)),
                                         variants=PROBE_VARIANTS,
                                         hard_limit_s=args.parity_probe_limit)
                    probe['shape'] = 'the timed workload\'s own'
                if probe is None or probe['ctc_loss_delta'] is None:
                    full = probe
                    probe = parity_probe(probe_kwargs, 'cuda:{}'.format(local_rank),
                                         variants=PROBE_VARIANTS)
                    probe['shape'] = 'small (batch 4, T\' = 100)'
                    if full is not None:
                        probe['full_size_attempt'] = full['note']
            except Exception as err:    # noqa: BLE001
                probe = {'ctc_loss_delta': None, 'logits_max_abs_delta': None,
                         'note': 'probe failed: {}: {}'.format(type(err).__name__, err)}
            result['ctc_loss_delta'] = probe['ctc_loss_delta']
            result['logits_max_abs_delta'] = probe['logits_max_abs_delta']
            result['parity_probe'] = probe
            # (an oracle child that could not run - a host without the time for it - leaves the
            # deltas null and says why; only a MEASURED delta above the bar fails the run: any
            # variant's, since every one of them is a number this line quotes)
            for variant in (probe.get('variants') or {'default': probe}).values():
                if variant['ctc_loss_delta'] is not None and (
                        variant['ctc_loss_delta'] > 1e-3 * max(1.0, abs(probe.get('loss_oracle')
                                                                       or 1.0)) or
                        variant['logits_max_abs_delta'] > 1e-3):
                    exit_code = 3                    # the line is printed, the run fails
        if world == 1 and not args.no_cpu_baseline and args.workload != 'c5':
            result['cpu_baseline'] = cpu_baseline(workload_cfg_kwargs(args.workload), batch,
                                                  seconds, frames)
            if result['cpu_baseline']['value']:
                result['gpu_over_cpu'] = round(result['value'] / result['cpu_baseline']['value'],
                                               1)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.stderr.write('bench.py: {}.\n'.format(
            'parity probe above the 1e-3 bar (see parity_probe)' if exit_code == 3 else
            'every release mode reported an error (see allreduce.modes)'))
        sys.exit(exit_code)


if __name__ == '__main__':
    main()
