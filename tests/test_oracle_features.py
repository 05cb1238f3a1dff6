"""Cross-checks of the feature-pipeline oracle (python_speech_features 0.6 restated) against
independent numpy / scipy formulations and closed-form cases."""

import numpy as np
import pytest
import scipy.fftpack
import scipy.signal

from oracle import features as ofeat


def _pcm(seconds, seed=0, rate=16000):
    rng = np.random.default_rng(seed)
    return np.clip(rng.normal(size=int(seconds * rate)) * 3000, -32768, 32767).astype(np.int16)


def test_frame_counts_of_the_baseline_configs():
    # 0.7 s -> 69, 3 s -> 299, 10 s -> 999, 17 s -> 1699 frames
    for seconds, frames in [(0.7, 69), (3.0, 299), (10.0, 999), (17.0, 1699)]:
        assert ofeat.num_frames(int(seconds * 16000)) == frames
    assert ofeat.num_frames(400) == 1 and ofeat.num_frames(401) == 2


def test_preemphasis_and_framing():
    x = np.arange(1, 1001, dtype=np.int16)
    y = ofeat.preemphasis(x)
    assert y[0] == 1 and np.allclose(y[1:], x[1:] - 0.97 * x[:-1].astype(np.float64))
    assert np.allclose(y, scipy.signal.lfilter([1, -0.97], [1], x.astype(np.float64)))
    frames = ofeat.frame_signal(y)
    assert frames.shape == (ofeat.num_frames(1000), 400)
    assert np.allclose(frames[1, :10], y[160:170])
    assert np.allclose(frames[-1, -(4 * 160 + 400 - 1000):], 0.0)     # zero padded tail


def test_power_spectrum_of_a_pure_tone_peaks_at_its_bin():
    t = np.arange(400) / 16000.0
    frame = np.cos(2 * np.pi * 1000.0 * t)[None, :]
    spec = ofeat.power_spectrum(frame)
    assert spec.shape == (1, 513) and int(np.argmax(spec[0])) == 64     # 1000 Hz * 1024 / 16000
    # Parseval on the zero-padded rfft (one-sided: interior bins count twice)
    full = spec[0, 0] + spec[0, -1] + 2 * spec[0, 1:-1].sum()
    assert full == pytest.approx(np.sum(frame ** 2), rel=1e-9)


def test_mel_filterbank_shape_and_edges():
    bank = ofeat.mel_filterbank()
    assert bank.shape == (80, 513) and bank.min() >= 0.0 and bank.max() <= 1.0
    first = np.nonzero(bank[0])[0]
    assert first[0] >= int(np.floor(1025 * 64.0 / 16000))            # nothing below 64 Hz
    assert np.all(bank[:, 0] == 0.0)
    centres = [int(np.argmax(row)) for row in bank]
    assert centres == sorted(centres)
    mel = ofeat.hz_to_mel(np.array([64.0, 1000.0, 8000.0]))
    assert np.allclose(ofeat.mel_to_hz(mel), [64.0, 1000.0, 8000.0])


def test_dct_lifter_delta_against_scipy_and_closed_forms():
    rng = np.random.default_rng(1)
    x = rng.normal(size=(11, 80))
    assert np.allclose(ofeat.dct2_ortho(x, 40),
                       scipy.fftpack.dct(x, type=2, axis=1, norm='ortho')[:, :40])
    lifted = ofeat.lifter(np.ones((2, 40)))
    assert np.allclose(lifted[0], 1 + 11 * np.sin(np.pi * np.arange(40) / 22))
    ramp = np.arange(20, dtype=np.float64)[:, None] * np.array([[1.0, -2.0]])
    d = ofeat.delta(ramp, 2)
    assert np.allclose(d[2:-2], [[1.0, -2.0]] * 16)                   # slope of a linear ramp
    assert np.allclose(d[0], (1 * 1 + 2 * 2) / 10 * np.array([1.0, -2.0]))   # edge padding


def test_log_mel_and_mfcc_shapes_dtypes_and_normalisation():
    pcm = _pcm(1.0)
    mel, length = ofeat.load_sample_from_pcm(pcm, feature_type='mel',
                                             feature_normalization='local')
    assert mel.shape == (99, 80) and mel.dtype == np.float32 and int(length) == 99
    assert np.allclose(mel.mean(axis=0), 0.0, atol=1e-4)
    assert np.allclose(mel.std(axis=0), 1.0, atol=1e-3)
    mfcc, _ = ofeat.load_sample_from_pcm(pcm, feature_type='mfcc', feature_normalization='none')
    assert mfcc.shape == (99, 80)
    feat, energy = ofeat.filterbank_energies(pcm)
    assert np.allclose(mfcc[:, 0], np.log(energy).astype(np.float32))     # c0 <- log energy
    assert np.allclose(mfcc[:, 40:], ofeat.delta(mfcc[:, :40].astype(np.float64), 2), atol=1e-4)
    raw, _ = ofeat.load_sample_from_pcm(pcm, feature_type='mel', feature_normalization='none')
    assert np.allclose(raw, np.log(feat).astype(np.float32))
    half, half_len = ofeat.load_sample_from_pcm(pcm, feature_type='mel',
                                                feature_normalization='none',
                                                drop_every_second_frame=True)
    assert int(half_len) == 50 and np.array_equal(half, raw[::2])
    scalar, _ = ofeat.load_sample_from_pcm(pcm, feature_type='mel',
                                           feature_normalization='local_scalar')
    assert scalar.mean() == pytest.approx(0.0, abs=1e-4)
    with pytest.raises(RuntimeError):
        ofeat.load_sample_from_pcm(pcm[:400])
    with pytest.raises(ValueError):
        ofeat.load_sample_from_pcm(pcm, feature_type='fbank')


def test_int16_pcm_is_not_rescaled():
    pcm = _pcm(0.5, seed=3)
    raw_i16, _ = ofeat.load_sample_from_pcm(pcm, 16000, 'mel', 'none')
    raw_f, _ = ofeat.load_sample_from_pcm(pcm.astype(np.float64) / 32768.0, 16000, 'mel', 'none')
    # scaling the signal by 1/32768 shifts every log-energy by -2*ln(32768)
    assert np.allclose(raw_i16 - raw_f, 2 * np.log(32768.0), atol=1e-3)
