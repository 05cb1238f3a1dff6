"""Corpus layout -> input pipeline -> training / evaluation drivers, end to end on the GPU with
a tiny synthetic corpus written in the reference's CSV/WAV layout."""

import os

import numpy as np
import pytest
import torch

from ctc_asr_amd import input_functions, storage, synth
from ctc_asr_amd.params import FLAGS

pytestmark = pytest.mark.gpu


@pytest.fixture()
def corpus(tmp_path):
    FLAGS.reset()
    corpus_dir = str(tmp_path / 'corpus')
    rng = np.random.default_rng(5)
    durations = np.round(rng.uniform(0.7, 2.0, size=21), 2)
    for name, seed, count in (('train', 1, 21), ('dev', 2, 9), ('test', 3, 9)):
        synth.write_corpus(corpus_dir, str(tmp_path / (name + '.csv')), durations[:count],
                           seed=seed, chars_per_second=6.0, subdir=name)
    FLAGS.update(corpus_dir=corpus_dir, train_csv=str(tmp_path / 'train.csv'),
                 dev_csv=str(tmp_path / 'dev.csv'), test_csv=str(tmp_path / 'test.csv'),
                 train_dir=str(tmp_path / 'ckpt'), batch_size=4, num_buckets=3,
                 feature_type='mel', feature_normalization='local', used_model='ds2',
                 conv_filters=[4, 4], num_units_dense=32, num_layers_rnn=1, num_units_rnn=64,
                 rnn_cell='lstm', max_epochs=2, learning_rate=1e-3, beam_width=8,
                 log_frequency=2, random_seed=7, dense_dropout_rate=0.0)
    yield tmp_path
    FLAGS.reset()


def test_load_sample_matches_oracle(corpus):
    from oracle import features as ofeat
    from scipy.io import wavfile
    rows = input_functions.read_manifest(FLAGS.train_csv)
    assert len(rows) == 21            # header and the sacrificial last row are dropped
    path = os.path.join(FLAGS.corpus_dir, rows[3]['path'])
    feats, length = input_functions.load_sample(path)
    _, pcm = wavfile.read(path)
    ref, ref_len = ofeat.load_sample_from_pcm(pcm, 16000, 'mel', 'local')
    assert feats.dtype == np.float32 and int(length) == int(ref_len) == feats.shape[0]
    assert np.abs(feats - ref).max() < 1e-3
    with pytest.raises(ValueError):
        input_functions.load_sample(path, feature_type='fbank')
    with pytest.raises(ValueError):
        input_functions.load_sample(path + '.missing')


def test_batching_semantics(corpus):
    ordered = list(input_functions.input_fn_generator('train_batch', prefetch=0)())
    assert len(ordered) == 5          # 21 examples, batch 4, remainder dropped
    rows = input_functions.read_manifest(FLAGS.train_csv)
    assert ordered[0].features['label_plaintext'] == [r['label'] for r in rows[:4]]
    feats = ordered[0].features['spectrogram']
    lens = ordered[0].features['spectrogram_length']
    assert feats.shape[0] == 4 and feats.shape[2] == 80 and feats.shape[1] == int(lens.max())
    short = int(torch.argmin(lens))
    assert float(feats[short, int(lens[short]):].abs().max()) == 0.0       # zero padding
    assert ordered[0].labels.dtype == np.int32 and ordered[0].labels.min() >= 0
    bucketed = list(input_functions.input_fn_generator('train_bucket', seed=3)())
    seen = sorted(t for b in bucketed for t in b.features['label_plaintext'])
    assert seen == sorted(r['label'] for r in rows)      # partial batches are kept
    # two ranks split every group of 2 * batch_size from the same bucket
    r0 = list(input_functions.input_fn_generator('train_batch', rank=0, world_size=2,
                                                 prefetch=0)())
    r1 = list(input_functions.input_fn_generator('train_batch', rank=1, world_size=2,
                                                 prefetch=0)())
    assert len(r0) == len(r1) == 2
    assert r0[0].features['label_plaintext'] == [r['label'] for r in rows[:4]]
    assert r1[0].features['label_plaintext'] == [r['label'] for r in rows[4:8]]


def test_train_resume_and_evaluate(corpus, capsys):
    from ctc_asr_amd import evaluate, train
    assert train.main([]) == 0
    out = capsys.readouterr().out
    assert 'Starting epoch 1 on train_batch' in out and 'Starting epoch 2 on train_bucket' in out
    assert 'examples/sec' in out and 'sec/batch' in out and 'audio-s/s' in out
    assert 'Completed all epochs.' in out
    ckpts = storage.checkpoint_paths(FLAGS.train_dir)
    assert len(ckpts) == 2
    # resume: nothing left to do, the latest checkpoint is picked up
    FLAGS.max_epochs = 3
    assert train.main([]) == 0
    out = capsys.readouterr().out
    assert 'Restored' in out and 'Starting epoch 3' in out and 'Starting epoch 1' not in out
    assert evaluate.main(['--dev']) == 0
    out = capsys.readouterr().out
    assert 'word_error_rate' in out and 'mean_edit_distance' in out
    # summaries: what the reference records with tf.summary / eval_metric_ops, as JSON lines
    from ctc_asr_amd import summaries
    train_records = summaries.read_summaries(FLAGS.train_dir, 'train')
    tags = {r['tag'] for r in train_records}
    assert {'loss', 'learning_rate', 'Metrics/mean_edit_distance', 'Metrics/word_error_rate',
            'audio_seconds_per_sec', 'decoded_text'} <= tags
    text = [r for r in train_records if r['tag'] == 'decoded_text'][0]['text']
    assert len(text) == 2 and 1 <= len(text[0]) <= FLAGS.num_samples_to_report
    dev = summaries.read_summaries(FLAGS.train_dir, 'eval_dev')
    assert {r['tag'] for r in dev} == {'loss', 'mean_edit_distance', 'word_error_rate'}
    assert len(dev) == 3 * 4              # 3 epochs by train.main + one evaluate.main --dev
    # --delete starts from scratch
    FLAGS.max_epochs = 1
    assert train.main(['--delete']) == 0
    assert len(storage.checkpoint_paths(FLAGS.train_dir)) == 1


def test_loss_decreases_on_a_fixed_batch(corpus):
    from ctc_asr_amd.engine import Trainer
    from ctc_asr_amd.model import ModelConfig
    batch = next(iter(input_functions.input_fn_generator('train_batch', prefetch=0)()))
    trainer = Trainer(ModelConfig.from_flags(FLAGS), flags=FLAGS, device='cuda', seed=1)
    feats, lens = batch.features['spectrogram'], batch.features['spectrogram_length']
    losses = [float(trainer.train_step(feats, lens, batch.labels)) for _ in range(30)]
    assert losses[-1] < losses[0] * 0.8


def test_model_fn_modes(corpus):
    from ctc_asr_amd.model import CTCModel, ModelConfig
    model = CTCModel(ModelConfig.from_flags(FLAGS), 'cuda', seed=3)
    batch = next(iter(input_functions.input_fn_generator('dev', seed=1, prefetch=0)()))
    features, labels = batch
    pred = model.model_fn(features, None, 'infer')
    assert set(pred) == {'decoded', 'plaintext'} and len(pred['decoded']) == len(labels)
    before = model.arena.param.clone()
    ev = model.model_fn(features, labels, 'eval')
    assert torch.equal(before, model.arena.param) and 0.0 <= float(ev['word_error_rate'])
    tr = model.model_fn(features, labels, 'train', learning_rate=1e-3)
    assert not torch.equal(before, model.arena.param) and torch.isfinite(tr['loss'])
    with pytest.raises(RuntimeError):
        model.model_fn(features, labels, 'tune')


def test_predict_main_transcribes_one_file(corpus, capsys):
    """`predict.main --input file.wav` (asr/predict.py:44-67): restores the latest checkpoint and
    prints {'decoded', 'plaintext'}; the transcription equals a beam decode of the same logits
    through the model API.  Missing file -> ValueError like the reference."""
    from ctc_asr_amd import predict, train
    from ctc_asr_amd.labels import decode
    from ctc_asr_amd.model import CTCModel, ModelConfig
    FLAGS.max_epochs = 1
    assert train.main([]) == 0
    capsys.readouterr()
    rows = input_functions.read_manifest(FLAGS.test_csv)
    wav = os.path.join(FLAGS.corpus_dir, rows[2]['path'])
    assert predict.main(['--input', wav]) == 0
    out = capsys.readouterr().out
    assert 'Inputs: ' + wav in out and "'plaintext'" in out and "'decoded'" in out
    model = CTCModel(ModelConfig.from_flags(FLAGS), 'cuda', seed=1)
    storage.restore_checkpoint(storage.latest_checkpoint(FLAGS.train_dir), model)
    got = predict.predict(model, wav)
    assert got['decoded'].dtype == np.int32
    assert got['plaintext'] == decode(got['decoded'].tolist())
    assert repr(got['plaintext']) in out
    feats, lengths = input_functions.features_from_pcm([input_functions.read_wav(wav)],
                                                       model.device)
    logits, seq_len = model.inference_fn(feats, lengths, training=False)
    decoded, _, _ = model.decode_fn(logits, seq_len, None)
    assert decoded[0] == got['decoded'].tolist()
    with pytest.raises(ValueError):
        predict.main(['--input', wav + '.missing'])


def test_the_ds2_model_memorises_a_small_batch():
    """End to end through the production kernels (own convolutions, persistent BiLSTM-1024
    forward / backward in step ranges, CTC, Adam): 400 steps on 8 noise utterances with random
    transcripts bring the loss from ~280 to < 0.5 and both decoders return every transcript."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        'overfit_check', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                      'tools', 'overfit_check.py'))
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    losses, greedy_hits, beam_hits, batch = module.run(steps=400, verbose=False)
    assert losses[0] > 100.0 and losses[-1] < 0.5, losses
    assert greedy_hits == batch and beam_hits == batch


def test_deferred_checks_still_raise_where_tensorflow_raises():
    """`Trainer.train_step(check=True)` does not stall the host on the CTC status words any more
    (ADVICE r02): an infeasible alignment - more labels than frames, where ``tf.nn.ctc_loss``
    raises InvalidArgumentError (asr/model.py:259) - surfaces at most `max_steps_ahead` steps
    later or at `drain_checks()`, names the step, and never passes a drain (= never reaches a
    checkpoint)."""
    from ctc_asr_amd.engine import Trainer
    from ctc_asr_amd.model import InfeasibleAlignmentError, ModelConfig
    cfg = ModelConfig(used_model='ds2', conv_filters=(4, 4), num_units_dense=32, num_layers_rnn=1,
                      num_units_rnn=64, rnn_cell='lstm', cudnn=True, dense_dropout_rate=0.0)
    trainer = Trainer(cfg, device='cuda', seed=3)
    rng = np.random.default_rng(3)
    feats = torch.tensor(rng.normal(size=(2, 21, 80)).astype(np.float32))      # T' = 11
    flen = torch.tensor([21, 21], dtype=torch.int32)
    good = [[1, 2, 3], [4, 5]]
    bad = [[1, 2, 3], list(range(1, 14))]                                      # 13 labels > 11
    trainer.train_step(feats, flen, good)
    trainer.drain_checks()
    before = trainer.model.arena.param.clone()
    moments = trainer.model.arena.m.clone(), trainer.model.arena.v.clone()
    trainer.train_step(feats, flen, bad)            # does not raise here: the check is deferred
    # ... but its update never reached the parameters or the moments: the Adam launch dropped it
    # on the device (nothing is applied from a step that is known to be invalid)
    torch.cuda.synchronize()
    assert torch.equal(trainer.model.arena.param, before)
    assert torch.equal(trainer.model.arena.m, moments[0])
    assert torch.equal(trainer.model.arena.v, moments[1])
    with pytest.raises(InfeasibleAlignmentError, match=r'batch rows \[1\].*training step 2'):
        trainer.drain_checks()
    trainer.drain_checks()                          # reported once
    trainer.train_step(feats, flen, good)           # a valid step moves them again
    torch.cuda.synchronize()
    assert not torch.equal(trainer.model.arena.param, before)
    trainer.drain_checks()
    # without a drain the error surfaces from a later train_step, within max_steps_ahead + 1
    trainer.train_step(feats, flen, bad)
    with pytest.raises(InfeasibleAlignmentError):
        for _ in range(trainer.max_steps_ahead + 1):
            trainer.train_step(feats, flen, good)
            torch.cuda.synchronize()


def test_a_nan_loss_raises_what_the_reference_raises():
    """A non-finite loss stops training with `NanLossDuringTrainingError` - the reference's
    NanTensorHook (asr/model.py:368) - from the Trainer's deferred checks as well (the
    guard word used to surface as a bare FloatingPointError before `train_epoch`'s own test of the
    loss could run); the step's update is dropped on the device."""
    from ctc_asr_amd import train
    from ctc_asr_amd.engine import NanLossDuringTrainingError, Trainer
    from ctc_asr_amd.model import ModelConfig
    assert train.NanLossDuringTrainingError is NanLossDuringTrainingError
    assert issubclass(NanLossDuringTrainingError, RuntimeError)
    cfg = ModelConfig(used_model='ds2', conv_filters=(4, 4), num_units_dense=32, num_layers_rnn=1,
                      num_units_rnn=64, rnn_cell='lstm', cudnn=True, dense_dropout_rate=0.0)
    trainer = Trainer(cfg, device='cuda', seed=3)
    rng = np.random.default_rng(5)
    feats = torch.tensor(rng.normal(size=(2, 21, 80)).astype(np.float32))
    flen = torch.tensor([21, 21], dtype=torch.int32)
    labels = [[1, 2, 3], [4, 5]]
    trainer.train_step(feats, flen, labels)
    trainer.drain_checks()
    before = trainer.model.arena.param.clone()
    poisoned = feats.clone()
    poisoned[1, 7, 3] = float('nan')
    trainer.train_step(poisoned, flen, labels)
    torch.cuda.synchronize()
    assert torch.equal(trainer.model.arena.param, before)
    with pytest.raises(NanLossDuringTrainingError, match='NaN loss during training'):
        trainer.drain_checks()
    # the dropped step leaves the step counter again (TensorFlow's global step would not have
    # advanced): Adam's bias correction and checkpoint names count applied updates
    with pytest.warns(RuntimeWarning, match='dropped on the device'):
        trainer.drain_checks()
    assert trainer.model.step_count == 1 and trainer.skipped_step_count() == 1
    trainer.train_step(feats, flen, labels)
    trainer.drain_checks()
    assert trainer.model.step_count == 2
    assert torch.isfinite(trainer.model.arena.param).all()


def test_step_guard_words():
    """`ctcasr_step_guard`: the skip word is set by a CTC status word, by a non-finite loss and by
    a recurrence time-out word, and only then; `adam_step(skip=)` honours it."""
    from ctc_asr_amd import hip
    hip.load()
    status = torch.zeros(5, dtype=torch.int32, device='cuda')
    loss = torch.ones(5, device='cuda')
    word = torch.zeros(1, dtype=torch.int32, device='cuda')
    assert hip.step_guard(status, loss).tolist() == [0, 0]
    assert hip.step_guard(status, loss, (word.data_ptr(), 0)).tolist() == [0, 0]
    word.fill_(1)
    assert hip.step_guard(status, loss, (0, word.data_ptr())).tolist() == [1, 1]
    status[3] = 1
    assert hip.step_guard(status, loss).tolist() == [1, 0]
    status.zero_()
    loss[4] = float('inf')
    assert hip.step_guard(status, loss).tolist() == [1, 0]
    loss[4] = float('nan')
    assert hip.step_guard(status, loss).tolist() == [1, 0]
    p, g = torch.ones(1000, device='cuda'), torch.ones(1000, device='cuda')
    m, v = torch.zeros(1000, device='cuda'), torch.zeros(1000, device='cuda')
    hip.adam_step(p, g, m, v, 1, lr=0.1, skip=torch.tensor([1, 0], dtype=torch.int32, device='cuda'))
    assert float(p.min()) == 1.0 and float(m.abs().max()) == 0.0
    hip.adam_step(p, g, m, v, 1, lr=0.1, skip=torch.tensor([0, 7], dtype=torch.int32, device='cuda'))
    assert float(p.max()) < 1.0 and float(m.min()) > 0.0


def test_out_of_range_weights_take_the_bf16_form(monkeypatch):
    """The fp16 form of the projection GEMMs scales W_ih by a FIXED 2^11 (|w| < 29).  Nothing
    bounds a weight: the model finds every matrix's largest magnitude on the device and a matrix
    outside half of that range takes the bf16 form (fp32's exponent range) - it never becomes
    inf.  One weight of 40 in the second layer: that layer's projections report 'bf16x6', the
    loss is finite and equals the all-fp32-GEMM step's; the fp16 split kernel itself saturates."""
    from ctc_asr_amd import hip, split_gemm
    from ctc_asr_amd.model import CTCModel, ModelConfig, init_params
    cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=2048,
                      num_layers_rnn=2, num_units_rnn=1024, rnn_cell='lstm', cudnn=True,
                      dense_dropout_rate=0.0)
    flat = init_params(cfg, 4)
    flat['rnn1/w_ih'][1, 17, 5] = 40.0
    rng = np.random.default_rng(4)
    feats = torch.tensor(rng.normal(size=(8, 199, 80)).astype(np.float32))
    flen = torch.full((8,), 199, dtype=torch.int32)
    labels = [list(rng.integers(1, 28, size=20)) for _ in range(8)]
    model = CTCModel(cfg, 'cuda', params=flat)
    loss = float(model.forward_backward(feats, flen, labels))
    what = model.arithmetic()
    assert what['rnn0/input_projection'] == 'fp16x3' and what['rnn1/input_projection'] == 'bf16x6'
    assert np.isfinite(loss) and torch.isfinite(model.arena.grad).all()
    monkeypatch.setenv('CTCASR_SPLIT_GEMM', '0')
    plain = CTCModel(cfg, 'cuda', params=flat)
    assert not plain.split_gemm
    loss_plain = float(plain.forward_backward(feats, flen, labels))
    assert abs(loss - loss_plain) <= 2e-6 * abs(loss_plain)
    rel = float((model.arena.g['rnn1/w_ih'] - plain.arena.g['rnn1/w_ih']).norm() /
                plain.arena.g['rnn1/w_ih'].norm())
    assert rel < 5e-4, rel
    # a weight that GROWS out of range is noticed from the maxima of an earlier step
    model.arena.p['rnn0/w_ih'][0, 3, 3] = 100.0
    for _ in range(4):
        model.forward_backward(feats, flen, labels)
        torch.cuda.synchronize()
    assert model.arithmetic()['rnn0/input_projection'] == 'bf16x6'
    # the split kernel saturates instead of producing inf
    x = torch.tensor([[1e3, -1e3, 40.0, 1.0, 0.0, 0.0, 0.0, 0.0]], device='cuda')
    pieces = hip.split_f16(x, split_gemm.W_SCALE, split_gemm.H_B)
    assert torch.isfinite(pieces.float()).all()
    assert float(pieces[0, 0, 0]) == 65504.0 and float(pieces[0, 0, 1]) == -65504.0
