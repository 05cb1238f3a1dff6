"""bench.py's launcher logic on a GPU-less host: `--gpus N` must never silently run a smaller
job (round-1 finding: it ran dp1 and labelled nothing)."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    full_env = {k: v for k, v in os.environ.items()
                if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'LOCAL_WORLD_SIZE')}
    full_env.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args,
                          capture_output=True, text=True, env=full_env, timeout=300)


def test_more_gpus_than_visible_is_refused_with_a_clear_message():
    import torch
    visible = torch.cuda.device_count() if torch.cuda.is_available() else 0
    out = _run(['--gpus', str(visible + 1 if visible else 2), '--steps', '1', '--warmup', '0'])
    assert out.returncode != 0
    assert 'GPU(s) visible' in out.stderr
    assert out.stdout.strip() == ''          # no JSON line that could be mistaken for a result


def test_world_size_mismatch_is_refused():
    out = _run(['--gpus', '4'], env={'WORLD_SIZE': '2', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert out.returncode != 0
    assert 'does not match WORLD_SIZE' in (out.stderr + out.stdout)


def test_pmc_traffic_lookup_reports_the_reason_when_absent(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.setattr(bench, 'PMC_TRAFFIC_JSON', str(tmp_path / 'none.json'))
    got = bench.pmc_traffic('c3', 'rnn_bwd', 167)
    assert got['traffic'] is None and 'pmc_traffic.json' in got['traffic_note']
    path = tmp_path / 'pmc.json'
    path.write_text(json.dumps({'entries': [
        {'workload': 'c3', 'pass': 'rnn_bwd', 'steps_per_launch': 167, 'fetch_kb': 1000.0,
         'write_kb': 500.0, 'algorithmic_bytes': 1024000, 'source': 'x.md'}]}))
    monkeypatch.setattr(bench, 'PMC_TRAFFIC_JSON', str(path))
    got = bench.pmc_traffic('c3', 'rnn_bwd', 167)
    assert got['traffic_raw'] == 1500 * 1024 and got['traffic'] == 2500 * 1024
    assert got['traffic_over_algorithmic'] == 2.5
    assert bench.pmc_traffic('c3', 'rnn_bwd', 500)['traffic'] is None


def test_release_modes_are_folded_into_one_line():
    """N > 1: bench.py measures the held and the early release of the gradient buckets and the
    stubbed step back to back; the printed line is the better mode's and carries all three."""
    sys.path.insert(0, ROOT)
    import bench

    def leg(mode, ms, value):
        return {'ms_per_step': ms, 'value': value, 'allreduce': {
            'mode': mode, 'launches_per_step': 9.0,
            'rank_ms_per_step': {'min': ms - 0.5, 'max': ms, 'all': [ms - 0.5, ms]}}}

    line = bench.merge_release_modes({'held': leg('held', 101.0, 6336.6),
                                      'early': leg('early', 99.5, 6432.2)},
                                     {'ms_per_step': 98.25})
    assert line['value'] == 6432.2 and line['allreduce']['chosen'] == 'early'
    modes = line['allreduce']['modes']
    assert set(modes) == {'held', 'early'}
    assert modes['held']['exposed_allreduce_ms'] == 2.75
    assert modes['early']['exposed_allreduce_ms'] == 1.25
    assert line['allreduce']['exposed_allreduce_ms'] == 1.25
    assert line['allreduce']['stubbed_ms_per_step'] == 98.25
    assert modes['early']['rank_ms_per_step']['min'] == 99.0
    # a mode that reported an error (a recurrence time-out on some rank) is never the chosen one
    bad = leg('early', 90.0, 7000.0)
    bad['allreduce']['failed'] = '1 rank(s) reported an error; rank 3: time-out'
    line = bench.merge_release_modes({'held': leg('held', 101.0, 6336.6), 'early': bad},
                                     {'ms_per_step': 98.25})
    assert line['allreduce']['chosen'] == 'held' and line['value'] == 6336.6
    assert line['allreduce']['modes']['early']['failed'].startswith('1 rank')
    assert 'failed' not in line['allreduce']


def test_c5_bucket_sequence_is_fixed_and_bucketed():
    """The C5 workload's batches: seeded (the same sequence every run), every batch from ONE
    bucket (its utterances differ by well under a second), lengths inside the corpus filter."""
    sys.path.insert(0, ROOT)
    import bench
    first = bench.c5_bucket_sequence(16, 12)
    again = bench.c5_bucket_sequence(16, 12)
    assert len(first) == 12 and all((a == b).all() for a, b in zip(first, again))
    for samples in first:
        assert samples.shape == (16,) and samples.min() >= 0.7 * 16000
        assert samples.max() <= 17.0 * 16000
        assert (samples.max() - samples.min()) / 16000.0 < 1.0
    longest = [int(s.max()) for s in first]
    assert max(longest) > 2 * min(longest)           # the padded length really varies


def test_flops_by_pipe_add_up_and_price_the_mixed_roof():
    import bench
    from ctc_asr_amd.model import ModelConfig
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=2048,
                      num_layers_rnn=5, num_units_rnn=1024, rnn_cell='lstm', cudnn=True)
    split, fp32 = bench.training_flops_by_pipe(cfg, 999)
    assert abs(split + fp32 - 3.0 * bench.forward_flops_per_utt(cfg, 999)) < 1e-3 * (split + fp32)
    assert split > 2.0 * fp32 > 0.0         # C3: the projections dominate
    with_split = bench.mixed_roof(cfg, 999, 32, 86.0, True)
    without = bench.mixed_roof(cfg, 999, 32, 96.0, False)
    # same FLOPs; 6 bf16 products at 2500 TF are 2.65 x the fp32 pipe's rate
    assert with_split['ms_per_step_at_peak'] < without['ms_per_step_at_peak']
    assert abs(without['ms_per_step_at_peak'] - 32 * (split + fp32) / 157.3e12 * 1e3) < 0.01
    assert 0.0 < with_split['frac'] < 1.0 and 0.0 < without['frac'] < 1.0


def test_predicted_input_bounds_and_dtype_label():
    """Host logic of round 4: which layers may take the fp16 form (bounded input), and the
    `dtype` text of the bench line for each arithmetic."""
    import types
    import bench
    from ctc_asr_amd import split_gemm
    from ctc_asr_amd.model import ModelConfig, predicted_input_bounds
    ds2 = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_layers_rnn=3, rnn_cell='lstm')
    assert predicted_input_bounds(ds2, True) == [20.0, 1.0, 1.0, 1.0]
    assert all(split_gemm.f16_scale(b) is not None for b in predicted_input_bounds(ds2, True))
    assert split_gemm.f16_scale(1.0) == split_gemm.RNN_F16_H_SCALE
    relu = ModelConfig(used_model='ds2', num_layers_rnn=2, rnn_cell='rnn_relu')
    assert predicted_input_bounds(relu, True) == [20.0, None, None]
    drop = ModelConfig(used_model='ds2', num_layers_rnn=2, rnn_cell='gru', rnn_dropout_rate=0.5,
                       conv_dropout_rate=0.2)
    assert predicted_input_bounds(drop, True) == [25.0, 2.0, 1.0]      # cuDNN: input of layers 2..L
    assert predicted_input_bounds(drop, False) == [25.0, 1.0, 1.0]
    ds1 = ModelConfig(used_model='ds1', num_layers_rnn=1, rnn_cell='lstm', cudnn=False,
                      dense_dropout_rate=0.1, rnn_dropout_rate=0.5, relu_cutoff=100.0)
    train_bounds = predicted_input_bounds(ds1, True)
    assert abs(train_bounds[0] - 100.0 / 0.9 / 0.5) < 1e-9 and train_bounds[1] == 2.0
    assert split_gemm.f16_scale(train_bounds[0]) is None               # cutoff above 64: bf16 form

    def model(**attrs):
        base = dict(split_gemm=True, fwd_f16=True, bwd_f16=True, rnn_fwd_f16=True, rnn_bwd_f16=True)
        base.update(attrs)
        return types.SimpleNamespace(**base)
    assert bench.dtype_label(model(split_gemm=False, rnn_fwd_f16=False, rnn_bwd_f16=False)) == 'f32'
    assert 'recurrence' in bench.dtype_label(model(split_gemm=False))
    text = bench.dtype_label(model())
    assert 'fp16x3' in text and 'forward and backward recurrence' in text and text.startswith('f32 ')
    assert 'recurrence' not in bench.dtype_label(model(rnn_fwd_f16=False, rnn_bwd_f16=False))
    assert 'fp16x3' not in bench.dtype_label(model(fwd_f16=False, rnn_fwd_f16=False,
                                                   rnn_bwd_f16=False))
