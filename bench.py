#!/usr/bin/env python
"""Headline benchmark: audio-seconds/s of CTC acoustic-model *training* on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one full training pass over one synthetic minibatch of raw 16 kHz int16 PCM that is
already resident in HBM: log-mel features + normalisation -> forward (conv front-end, BiLSTM
stack, dense, logits) -> fused log-softmax + CTC loss/gradient -> backward -> RCCL gradient
all-reduce (N > 1) -> TensorFlow-form Adam.  Workload =
BASELINE.json configs[1]: DS2, 2 conv layers + 2 x BiLSTM-1024, batch 16 per GPU, 10 s
utterances (999 feature frames -> T' = 500), fp32.  Scaling is weak: per-GPU batch fixed.

Rank 0 prints ONE JSON line (schema in the task contract) that also carries
  "roofline":     the dominant kernel (the recurrent time-step kernel) against its roof,
                  from HIP events recorded live around every recurrence call in the timed
                  region, and
  "cpu_baseline": the same graph in stock torch CPU operators (oracle/torch_ref.py, kind
                  "port" - the reference's TensorFlow cannot run offline) on the host cores,
                  rank 0, N = 1 only, bounded sample.
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (conv_filters, rnn layers, hidden, dense, per-GPU batch, seconds, rnn_cell)
    'c2': ((32, 32), 2, 1024, 2048, 16, 10.0, 'lstm'),           # BASELINE.json configs[1] (metric)
    'c2_3conv': ((32, 32, 96), 2, 1024, 2048, 16, 10.0, 'lstm'),  # with the reference's 3 convs
    'c3': ((32, 32), 5, 1024, 2048, 32, 10.0, 'lstm'),            # configs[2]
    'c3_3conv': ((32, 32, 96), 5, 1024, 2048, 32, 10.0, 'lstm'),
    # the reference's own flag defaults (asr/params.py): 3 convs, 4 x ReLU-RNN-2048, batch 16
    'ref_default': ((32, 32, 96), 4, 2048, 2048, 16, 10.0, 'rnn_relu'),
    'tiny': ((8, 8), 1, 128, 128, 4, 2.0, 'lstm'),                # plumbing check
}

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md chip table
# (FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch, measured with rocprofv3 PMC passes
# (key: workload, pass, time steps per launch)
PMC_TRAFFIC = {('c2', 'rnn_bwd', 167): int((486779 + 171415) * 1024),
               ('c2', 'rnn_bwd', 500): int((1426952 + 513996) * 1024),
               ('c2', 'rnn_fwd', 500): int((793441 + 451992) * 1024)}
HBM_PEAK_GBS = 8000.0


def forward_flops_per_utt(cfg, frames):
    """SURVEY.md 8d formula: conv + input/recurrent projections + dense4 + logits."""
    from ctc_asr_amd.model import CONV_KERNEL_SIZES, GATES
    t_out = cfg.output_time(frames)
    flops, c_in, freq = 0.0, 1, cfg.num_features
    if cfg.used_model == 'ds2':
        for i, c_out in enumerate(cfg.conv_filters):
            freq = -(-freq // 2)
            k_t, k_f = CONV_KERNEL_SIZES[i]
            flops += 2.0 * t_out * freq * c_out * k_t * k_f * c_in
            c_in = c_out
    gates, hidden = GATES[cfg.cell], cfg.num_units_rnn
    in_size = cfg.rnn_input_size()
    for _ in range(cfg.num_layers_rnn):
        flops += 2.0 * t_out * 2 * gates * hidden * (in_size + hidden)
        in_size = 2 * hidden
    flops += 2.0 * t_out * 2 * hidden * cfg.num_units_dense
    flops += 2.0 * t_out * cfg.num_units_dense * cfg.num_classes
    return flops


def _cpu_baseline_worker(spec):
    """Runs in a child process: time the torch-CPU restatement (fwd + bwd + TF-Adam)."""
    from ctc_asr_amd.model import ModelConfig, init_params, to_oracle_layout
    from ctc_asr_amd.synth import synthetic_batch
    from oracle import torch_ref
    cfg = ModelConfig(**spec['cfg'])
    threads, batch, seconds = spec['threads'], spec['batch'], spec['seconds']
    torch.set_num_threads(threads)
    feats, lengths, labels, _ = synthetic_batch(batch, seconds, seed=99, frames=spec['frames'])
    label_rows = [[int(v) for v in row if v] for row in labels]
    model = torch_ref.TorchRefModel(to_oracle_layout(init_params(cfg, 0), cfg), cfg.used_model,
                                    cfg.rnn_cell, cfg.cudnn)
    opt = torch_ref.TFAdam(model.parameters())
    feats_t = torch.tensor(feats)

    def one_step():
        opt.zero_grad()
        logits, seq_len = model(feats_t, lengths)
        loss, _ = model.loss(logits, seq_len, label_rows)
        loss.backward()
        opt.step()

    t0 = time.perf_counter()
    one_step()                               # warm-up (also tells us how long a step takes)
    warm = time.perf_counter() - t0
    steps = int(max(1, min(10, (spec['budget_s'] - warm) // max(warm, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(steps):
        one_step()
    per_step = (time.perf_counter() - t0) / steps
    print(json.dumps({'steps': steps, 'per_step': per_step, 'warm': warm}))


def cpu_baseline(cfg_kwargs, seconds, frames, budget_s=20.0, hard_limit_s=150.0):
    """CPU baseline (kind "port") on a bounded sample, in a child process with a hard time
    limit so that a slow host can never stall the benchmark."""
    import subprocess
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    batch = 2
    code = ('import json,sys; sys.path.insert(0, {!r}); import bench; '
            'bench._cpu_baseline_worker(json.loads(sys.argv[1]))').format(ROOT)
    # torch's CPU LSTM stops scaling well before all cores of a big host: time two thread counts
    # (half the budget each) and report the faster one with the thread count it used
    info, threads = None, 1
    for cand in sorted({max(1, min(cores, 16)), max(1, min(cores, 64))}):
        spec = {'cfg': cfg_kwargs, 'threads': cand, 'batch': batch, 'seconds': seconds,
                'frames': frames, 'budget_s': budget_s / 2}
        env = dict(os.environ, OMP_NUM_THREADS=str(cand), MKL_NUM_THREADS=str(cand))
        try:
            out = subprocess.run([sys.executable, '-c', code, json.dumps(spec)], env=env,
                                 capture_output=True, text=True, timeout=hard_limit_s / 2)
            got = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:      # timeout or failure of this candidate: never block the GPU result
            continue
        if info is None or got['per_step'] < info['per_step']:
            info, threads = got, cand
    if info is None:
        return {'value': None, 'unit': 'audio-s/s', 'cores': threads, 'kind': 'port',
                'sample': 'CPU baseline did not finish within {:.0f} s'.format(hard_limit_s)}
    return {'value': round(batch * seconds / info['per_step'], 3), 'unit': 'audio-s/s',
            'cores': threads, 'kind': 'port',
            'sample': '{} timed fwd+bwd+Adam steps after 1 warm-up, batch {} x {:.0f} s, torch '
                      '{} CPU ops (oracle/torch_ref.py), {} of {} host cores, {:.2f} s/step'
                      .format(info['steps'], batch, seconds, torch.__version__, threads, cores,
                              info['per_step'])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='c2', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dropout', type=float, default=0.1,
                    help='dense_dropout_rate (reference default 0.1)')
    ap.add_argument('--rnn-bwd-whole-chip', action='store_true',
                    help='persistent backward recurrence on all 256 CUs (default: 128)')
    args = ap.parse_args()

    from ctc_asr_amd import hip
    from ctc_asr_amd.engine import Trainer, init_distributed
    from ctc_asr_amd.model import CTCModel, GATES, ModelConfig
    from ctc_asr_amd.synth import synthetic_batch
    import torch.distributed as dist

    if args.rnn_bwd_whole_chip:
        hip.set_option('rnn_bwd_half_chip', 0)
    rank, local_rank, world = init_distributed()
    if world != args.gpus and world > 1:
        raise SystemExit('--gpus {} does not match WORLD_SIZE {}'.format(args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X; no GPU is visible.')
    torch.cuda.set_device(local_rank)
    device = 'cuda:{}'.format(local_rank)

    filters, layers, hidden, dense, batch, seconds, rnn_cell = WORKLOADS[args.workload]
    cfg = ModelConfig(used_model='ds2', conv_filters=filters, num_units_dense=dense,
                      num_layers_rnn=layers, num_units_rnn=hidden, rnn_cell=rnn_cell, cudnn=True,
                      dense_dropout_rate=args.dropout)
    # fixed input shape: let MIOpen benchmark its convolution kernels once (warm-up steps)
    trainer = Trainer(cfg, device=device, seed=0, world_size=world, rank=rank, conv_autotune=True)
    model = trainer.model

    # synthetic 16 kHz utterances: int16 PCM resident in HBM (SURVEY.md 8d recipe), random labels
    from ctc_asr_amd.synth import random_pcm
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
        return trainer.train_step(feat_buf, len_d, packed, check=False)

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    CTCModel.check_status(model.last_status)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    hip.EVENTS = {}
    hip.set_option('rnn_kernel_events', 1)      # event pairs right around the persistent kernels
    hip.rnn_kernel_events()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    events = hip.drain_events()
    hip.EVENTS = None
    kernel_events = hip.rnn_kernel_events()
    hip.set_option('rnn_kernel_events', 0)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    CTCModel.check_status(model.last_status)
    # a persistent recurrence launch that gave up at a grid barrier would have produced garbage
    hip.rnn_poll_error(cfg.cell, model._acts['rnn_ws'], cfg.output_time(frames), batch, hidden)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        audio_s = world * batch * seconds
        value = audio_s / (elapsed / args.steps)
        t_out = cfg.output_time(frames)
        flops_per_audio_s = 3.0 * forward_flops_per_utt(cfg, frames) / seconds
        gates = GATES[cfg.cell]
        # dominant kernel: the recurrence.  Persistent variant: one launch per layer-pass runs all
        # T' steps with the recurrent weights resident in LDS -> fp32-MFMA roof (the limiter is the
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
                # the backward recurrence of a layer may be cut into several launches
                # (CTCModel.bwd_chunks); a launch then covers T' / chunks time steps
                launch_steps = t_out * args.steps * cfg.num_layers_rnn / float(calls)
                achieved = flops_per_step * launch_steps / avg_s / 1e12
                roofline = {
                    'kernel': 'prnn_{}_kernel<{}> (persistent, LDS-resident recurrent weights; '
                              'one launch = {:.0f} time steps x 2 directions)'.format(
                                  dom[4:], rnn_cell.upper(), launch_steps),
                    'bound': 'mfma', 'achieved': round(achieved, 2),
                    'peak': FP32_MFMA_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                    # HBM-side bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
                    # passes of this workload (profiles/r01_final_c2_kernel_trace_and_pmc.md)
                    'traffic': PMC_TRAFFIC.get((args.workload, dom, round(launch_steps))),
                    'avg_launch_us': round(avg_s * 1e6, 1), 'launches': calls,
                    'algorithmic_flops_per_launch': flops_per_step * launch_steps,
                    'us_per_time_step': round(avg_s * 1e6 / launch_steps, 3),
                    'share_of_step': round(dom_ms / (elapsed * 1e3), 3)}
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
            'dtype': 'f32', 'data': 'synthetic (16 kHz int16 Gaussian-noise PCM of fixed-length '
                                    'utterances resident in HBM, random labels at 15 chars/s)',
            'config': {'workload': '{}: DS2 {}-conv + {}xBi{}-{}, '
                                   'batch {}/GPU, {:.0f} s utterances'.format(
                                       {'c2': 'BASELINE.json configs[1]',
                                        'c3': 'BASELINE.json configs[2]'}.get(
                                            args.workload, args.workload), len(filters), layers,
                                       {'lstm': 'LSTM', 'rnn_relu': 'RNN(relu)'}[rnn_cell],
                                       hidden, batch, seconds),
                       'name': args.workload, 'global_batch': world * batch, 'frames': frames,
                       'ctc_steps': t_out, 'parallelism': 'dp{}'.format(world),
                       'dense_dropout_rate': args.dropout,
                       'parameters': model.arena.num_parameters()},
            'loss': round(float(loss), 4),
            'step_tflops_fp32': round(value * flops_per_audio_s / 1e12, 2),
            'frac_of_fp32_mfma_peak': round(value * flops_per_audio_s / 1e12 /
                                            (FP32_MFMA_PEAK_TFLOPS * world), 4),
            'kernel_ms_per_step': {k: round(v[1] / args.steps, 3) for k, v in events.items()},
            'roofline': roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            cfg_kwargs = dict(used_model='ds2', conv_filters=list(filters), num_units_dense=dense,
                              num_layers_rnn=layers, num_units_rnn=hidden, rnn_cell=rnn_cell,
                              cudnn=True, dense_dropout_rate=0.0)
            result['cpu_baseline'] = cpu_baseline(cfg_kwargs, seconds, frames)
            if result['cpu_baseline']['value']:
                result['gpu_over_cpu'] = round(value / result['cpu_baseline']['value'], 1)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
