"""Summary records and the LoggerHook-style throughput line (host only)."""

import re

from ctc_asr_amd import summaries


def test_writer_round_trip(tmp_path):
    writer = summaries.SummaryWriter(str(tmp_path), 'train')
    writer.scalar('loss', 12.5, 200)
    writer.scalar('Metrics/word_error_rate', 0.25, 200)
    writer.text('decoded_text', [['a b', 'c'], ['a d', 'c']], 200)
    records = summaries.read_summaries(str(tmp_path), 'train')
    assert [r['tag'] for r in records] == ['loss', 'Metrics/word_error_rate', 'decoded_text']
    assert records[0]['value'] == 12.5 and records[0]['step'] == 200
    assert records[2]['text'] == [['a b', 'c'], ['a d', 'c']]
    assert summaries.read_summaries(str(tmp_path), 'eval_dev') == []


def test_throughput_line_has_the_reference_fields():
    logger = summaries.ThroughputLogger(log_frequency=10, batch_size=16)
    logger.add_audio(1600.0)
    line, examples_per_sec, audio_per_sec = logger.line(1230, 3.14159)
    # asr/util/hooks.py:470-477: '(step=1,230); loss=3.1416; N examples/sec (S sec/batch) (B batch/sec)'
    assert re.search(r'\(step=1,230\); loss=3\.1416; [\d.]+ examples/sec \([\d.]+ sec/batch\) '
                     r'\([\d.]+ batch/sec\); [\d.]+ audio-s/s', line)
    assert examples_per_sec > 0 and audio_per_sec > 0
    _, _, audio_after = logger.line(1240, 1.0)
    assert audio_after == 0.0            # the window restarts


def test_throughput_counts_the_steps_actually_in_the_window(monkeypatch):
    """The first line of an epoch is logged after ONE step: examples/sec must be one batch over
    the window, not log_frequency batches (the reference's hook assumes the latter)."""
    now = [100.0]
    monkeypatch.setattr(summaries.time, 'time', lambda: now[0])
    logger = summaries.ThroughputLogger(log_frequency=200, batch_size=16)
    logger.add_audio(160.0)
    now[0] += 2.0
    _, examples_per_sec, audio_per_sec = logger.line(1, 5.0)
    assert examples_per_sec == 8.0 and audio_per_sec == 80.0
    now[0] += 30.0                       # decode + summaries of the log step: not training time
    logger.restart()
    for _ in range(200):
        logger.add_audio(160.0)
    now[0] += 50.0
    line, examples_per_sec, _ = logger.line(201, 4.0)
    assert examples_per_sec == 64.0 and '(0.250 sec/batch)' in line
