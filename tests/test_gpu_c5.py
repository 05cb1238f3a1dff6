"""BASELINE.json configs[4] (C5) on one GPU: mixed-length utterances (0.7 s ... 17 s, the
reference's corpus filter ``asr/params.py:142-143``) drawn as bucketed batches of 16 through
``input_fn_generator('train_bucket')`` (``asr/input_functions.py:84-98``) into DS2 2-conv +
2 x BiLSTM-1024 - i.e. through the persistent LDS-resident recurrence kernels at T' = 35 (the
shortest batch the corpus filter allows), an odd T' and T' = 850 (17 s) - with the beam-search
decode of width 64 the config names (``asr/model.py:292-296``).

Checked per batch: logits and loss against the float64 torch restatement fed the SAME features
(1e-3, the north_star bar); the beam-64 decode of the model's own logits identical to the C
oracle's on those logits; no recurrence time-out.  For the shortest and the odd-T' batch also
every gradient; for the longest a whole training step (backward in step ranges at T' = 850)."""

import numpy as np
import pytest
import torch

from ctc_asr_amd import input_functions, synth
from ctc_asr_amd.params import FLAGS
from oracle import cref, torch_ref

pytestmark = pytest.mark.gpu

BATCH = 16
SHORT, ODD_MAX, LONG = 0.7, 6.58, 17.0     # seconds -> 69 / 657 / 1699 frames -> T' 35 / 329 / 850


@pytest.fixture(scope='module')
def c5_corpus(tmp_path_factory):
    """49 utterances in the reference's CSV / WAV layout: 17 x 0.7 s, 16 in [6.0, 6.58] s (ragged
    inside the batch, longest -> odd T'), 16 in [15.5, 17.0] s; 15 chars per second of audio."""
    root = tmp_path_factory.mktemp('c5')
    rng = np.random.default_rng(17)
    durations = [SHORT] * 17
    durations += list(np.round(rng.uniform(6.0, 6.5, size=15), 2)) + [ODD_MAX]
    durations += list(np.round(rng.uniform(15.5, 16.9, size=15), 2)) + [LONG]
    FLAGS.reset()
    corpus_dir = str(root / 'corpus')
    synth.write_corpus(corpus_dir, str(root / 'train.csv'), durations, seed=3, subdir='train')
    FLAGS.update(corpus_dir=corpus_dir, train_csv=str(root / 'train.csv'), batch_size=BATCH,
                 num_buckets=3, feature_type='mel', feature_normalization='local',
                 used_model='ds2', conv_filters=[32, 32], num_units_dense=2048,
                 num_layers_rnn=2, num_units_rnn=1024, rnn_cell='lstm', beam_width=64,
                 dense_dropout_rate=0.0, random_seed=7, shuffle_buffer_size=64)
    yield root
    FLAGS.reset()


def _bucketed_batches():
    batches = list(input_functions.input_fn_generator('train_bucket', seed=5, prefetch=0)())
    by_steps = {}
    for batch in batches:
        feats = batch.features['spectrogram']
        if feats.shape[0] == BATCH:
            by_steps[(feats.shape[1] + 1) // 2] = batch
    return batches, by_steps


def _model():
    from ctc_asr_amd.model import CTCModel, ModelConfig, init_params
    cfg = ModelConfig.from_flags(FLAGS)
    flat = init_params(cfg, 31)
    rng = np.random.default_rng(31)
    for name in flat:       # non-zero biases, livelier weights than the tiny-sigma initialiser
        flat[name] = (flat[name] + rng.normal(size=flat[name].shape) * 0.01).astype(np.float32)
    return cfg, flat, CTCModel(cfg, 'cuda', params=flat)


def _reference(cfg, flat, feats, labels, with_grads):
    from ctc_asr_amd.model import to_oracle_layout
    threads = torch.get_num_threads()
    torch.set_num_threads(max(threads, 32))      # fp64 on the host: the long batch is ~1 TFLOP
    try:
        ref = torch_ref.TorchRefModel(to_oracle_layout(flat, cfg), 'ds2', 'lstm', True,
                                      dtype=torch.float64)
        x = feats.cpu().double()
        lengths = np.full(x.shape[0], x.shape[1], dtype=np.int32)
        with torch.set_grad_enabled(with_grads):
            logits, seq_len = ref(x, lengths)
            loss, _ = ref.loss(logits, seq_len, labels)
            if with_grads:
                loss.backward()
        return logits.detach().numpy(), float(loss), (ref.grads_in_shared_layout()
                                                      if with_grads else None)
    finally:
        torch.set_num_threads(threads)


def _label_rows(batch):
    return [[int(v) for v in row if v] for row in batch.labels]


def test_the_corpus_yields_the_c5_batches(c5_corpus):
    batches, by_steps = _bucketed_batches()
    # every example exactly once, partial final batches kept (asr/input_functions.py:96)
    assert sum(b.features['spectrogram'].shape[0] for b in batches) == 49
    assert {35, 329, 850} <= set(by_steps)
    mid = by_steps[329].features
    assert int(mid['spectrogram_length'].min()) < int(mid['spectrogram_length'].max()) == 657
    row = int(torch.argmin(mid['spectrogram_length']))
    # padding is zeros appended AFTER normalisation (asr/input_functions.py:112-120)
    assert float(mid['spectrogram'][row, int(mid['spectrogram_length'][row]):].abs().max()) == 0.0


@pytest.mark.parametrize('steps,with_grads', [(35, True), (329, True), (850, False)])
def test_bucketed_batch_through_the_persistent_bilstm(c5_corpus, steps, with_grads):
    from ctc_asr_amd import hip
    _, by_steps = _bucketed_batches()
    batch = by_steps[steps]
    feats = batch.features['spectrogram']
    labels = _label_rows(batch)
    assert hip.rnn_persistent_supported('lstm', steps, BATCH, 1024)
    cfg, flat, model = _model()
    logits, seq_len = model.inference_fn(feats, batch.features['spectrogram_length'],
                                         training=True)
    loss = model.loss_fn(logits, seq_len, batch.labels)
    model.backward()
    model.check_rnn_error()
    assert logits.shape == (steps, BATCH, 29)
    # DS2: every row's sequence length is the PADDED T' (asr/model.py:159-163 quirk)
    assert (seq_len.cpu().numpy() == steps).all()
    ref_logits, ref_loss, ref_grads = _reference(cfg, flat, feats, labels, with_grads)
    got = logits.cpu().numpy()
    assert np.abs(got - ref_logits).max() < 1e-3
    assert abs(float(loss) - ref_loss) < 1e-3 * max(1.0, abs(ref_loss))
    # beam search, width 64, on the model's own logits: identical paths to the C oracle.  (The
    # barely initialised network's logits are flat to ~1e-2, so competing beams sit within an
    # ulp of expf / log1pf of each other and glibc and the GPU's libm then order them differently
    # - DESIGN.md section 2; both decoders get the same sharpened copy.)
    sharp = (logits * 30.0).contiguous()
    decoded, plaintext, _ = model.decode_fn(sharp, seq_len, None, beam_width=64)
    want, _ = cref.beam_search_decode(sharp.cpu().numpy(), np.full(BATCH, steps, dtype=np.int32),
                                      64)
    assert decoded == want
    assert len(plaintext) == BATCH
    if with_grads:
        grads = model.arena.export('grad')
        pairs = [('conv0/kernel', ref_grads['conv'][0][0]), ('conv1/kernel', ref_grads['conv'][1][0]),
                 ('conv1/bias', ref_grads['conv'][1][1]),
                 ('dense4/kernel', ref_grads['dense4'][0]), ('logits/bias', ref_grads['logits'][1])]
        for i, layer in enumerate(ref_grads['rnn']):
            pairs += [('rnn{}/{}'.format(i, k), layer[k]) for k in ('w_ih', 'w_hh', 'b_ih', 'b_hh')]
        for name, want_g in pairs:
            want_g = want_g.numpy()
            err = np.abs(grads[name] - want_g).max()
            assert err < 1e-3 * max(1.0, np.abs(want_g).max()), (name, err)
    else:
        assert torch.isfinite(model.arena.grad).all()
        assert float(model.arena.g['rnn0/w_hh'].abs().max()) > 0.0


def test_training_over_the_bucket_sequence_and_deferred_decode(c5_corpus):
    """A pass over ALL bucketed batches (T' changes every step, a final partial batch of one
    utterance included) through `Trainer.train_step` with its deferred checks, then the dev-style
    evaluation path: logits of all batches decoded in ONE beam-64 launch (`decode_many`) equal
    to the C oracle batch by batch."""
    from ctc_asr_amd.engine import Trainer
    batches, _ = _bucketed_batches()
    # (livelier weights than the initialiser's: an untrained network's logits are flat to ~1e-3,
    # where competing beams differ by less than an ulp of expf / log1pf between libm and ocml)
    cfg, flat, _ = _model()
    trainer = Trainer(cfg, flags=FLAGS, device='cuda', params=flat)
    before = trainer.model.arena.param.clone()
    losses = []
    for _ in range(2):
        for batch in batches:
            losses.append(trainer.train_step(batch.features['spectrogram'],
                                             batch.features['spectrogram_length'], batch.labels))
    trainer.drain_checks()
    losses = [float(v) for v in losses]
    assert all(np.isfinite(losses)) and min(losses) > 0.0
    assert trainer.model.step_count == 2 * len(batches)
    model = trainer.model
    assert torch.isfinite(model.arena.param).all() and not torch.equal(before, model.arena.param)
    pending = []
    for batch in batches:
        logits, seq_len = model.inference_fn(batch.features['spectrogram'],
                                             batch.features['spectrogram_length'],
                                             training=False)
        # (sharpened: near-ties of a barely trained network are not a parity case, see above)
        pending.append(((logits * 30.0).contiguous(), seq_len.clone(), None))
    model.check_rnn_error()
    results = model.decode_many(pending, beam_width=64)
    for (logits, seq_len, _), (decoded, _, _) in zip(pending, results):
        want, _ = cref.beam_search_decode(logits.cpu().numpy(), seq_len.cpu().numpy(), 64)
        assert decoded == want
