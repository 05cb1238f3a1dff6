"""ORACLE (test infrastructure, not product code): CPU restatement of the feature pipeline.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  PARITY UNPINNED by the reference: it ships no tests or vectors for this path and
neither it nor ``python_speech_features`` (psf) can run in the build container.

What is restated, float64 throughout like psf/numpy:

* ``load_sample`` post-processing — ``asr/input_functions.py:156-250``
* ``__mel``  -> ``psf.logfbank`` — ``asr/input_functions.py:285-304``
* ``__mfcc`` -> ``psf.mfcc`` + ``psf.delta`` — ``asr/input_functions.py:253-282``
* ``__feature_normalization`` — ``asr/input_functions.py:307-335``

psf is python_speech_features 0.6 (``requirements.txt`` lower bound); its published algorithm
(sigproc.preemphasis / framesig / powspec, base.get_filterbanks / fbank / logfbank / mfcc /
lifter / delta) is followed step by step and cross-checked in ``tests/test_oracle_features.py``
against independent numpy/scipy formulations.
"""

import decimal
import math

import numpy as np

WIN_LENGTH = 0.025
WIN_STEP = 0.010
NUM_FEATURES = 80
N_FFT = 1024
F_MIN = 64.0
PRE_EMPHASIS = 0.97
CEP_LIFTER = 22
EPS = np.finfo(float).eps


def _round_half_up(number):
    return int(decimal.Decimal(number).quantize(decimal.Decimal('1'),
                                                rounding=decimal.ROUND_HALF_UP))


def num_frames(num_samples, sampling_rate=16000):
    """Frame count of psf ``framesig``: 1 + ceil((N - frame_len) / frame_step), min 1."""
    frame_len = _round_half_up(WIN_LENGTH * sampling_rate)
    frame_step = _round_half_up(WIN_STEP * sampling_rate)
    if num_samples <= frame_len:
        return 1
    return 1 + int(math.ceil((1.0 * num_samples - frame_len) / frame_step))


def preemphasis(signal, coeff=PRE_EMPHASIS):
    """y[0] = x[0]; y[n] = x[n] - coeff * x[n-1].  Raw int16 PCM is *not* rescaled
    (``asr/input_functions.py:209``: the array from ``wavfile.read`` goes straight in)."""
    signal = np.asarray(signal)
    out = np.empty(signal.shape[0], dtype=np.float64)
    out[0] = signal[0]
    out[1:] = signal[1:] - coeff * signal[:-1]
    return out


def frame_signal(signal, sampling_rate=16000):
    """Overlapping frames (len 400, step 160 at 16 kHz), zero padded tail, rectangular window."""
    frame_len = _round_half_up(WIN_LENGTH * sampling_rate)
    frame_step = _round_half_up(WIN_STEP * sampling_rate)
    count = num_frames(len(signal), sampling_rate)
    padded_len = (count - 1) * frame_step + frame_len
    padded = np.concatenate([signal, np.zeros(padded_len - len(signal))])
    index = np.arange(frame_len)[None, :] + frame_step * np.arange(count)[:, None]
    return padded[index]


def power_spectrum(frames, n_fft=N_FFT):
    """|rfft(frame, n_fft)|^2 / n_fft -> [T, n_fft/2 + 1]."""
    return np.square(np.absolute(np.fft.rfft(frames, n_fft))) / n_fft


def hz_to_mel(hz):
    return 2595.0 * np.log10(1.0 + hz / 700.0)


def mel_to_hz(mel):
    return 700.0 * (10.0 ** (mel / 2595.0) - 1.0)


def mel_filterbank(num_filters=NUM_FEATURES, n_fft=N_FFT, sampling_rate=16000,
                   low_freq=F_MIN, high_freq=None):
    """Triangular filters on floor((n_fft + 1) * hz / rate) bin edges -> [num_filters, n_fft/2+1]."""
    high_freq = high_freq or sampling_rate / 2
    mel_points = np.linspace(hz_to_mel(low_freq), hz_to_mel(high_freq), num_filters + 2)
    edges = np.floor((n_fft + 1) * mel_to_hz(mel_points) / sampling_rate)
    bank = np.zeros((num_filters, n_fft // 2 + 1))
    for j in range(num_filters):
        lo, mid, hi = edges[j], edges[j + 1], edges[j + 2]
        for i in range(int(lo), int(mid)):
            bank[j, i] = (i - lo) / (mid - lo)
        for i in range(int(mid), int(hi)):
            bank[j, i] = (hi - i) / (hi - mid)
    return bank


def filterbank_energies(signal, sampling_rate=16000, num_filters=NUM_FEATURES, n_fft=N_FFT,
                        low_freq=F_MIN, high_freq=None):
    """psf ``fbank``: (mel energies [T, nfilt], frame energy [T]); exact zeros become eps."""
    frames = frame_signal(preemphasis(signal), sampling_rate)
    pspec = power_spectrum(frames, n_fft)
    energy = np.sum(pspec, axis=1)
    energy = np.where(energy == 0, EPS, energy)
    feat = pspec @ mel_filterbank(num_filters, n_fft, sampling_rate, low_freq, high_freq).T
    feat = np.where(feat == 0, EPS, feat)
    return feat, energy


def log_mel(signal, sampling_rate=16000, num_features=NUM_FEATURES):
    """``__mel``: psf.logfbank(nfilt=80, nfft=1024, lowfreq=64, highfreq=rate/2, preemph=.97)."""
    feat, _ = filterbank_energies(signal, sampling_rate, num_features, N_FFT, F_MIN,
                                  sampling_rate / 2.0)
    return np.log(feat)


def dct2_ortho(x, num_out):
    """scipy.fftpack.dct(x, type=2, axis=1, norm='ortho')[:, :num_out] as an explicit matrix."""
    n = x.shape[1]
    k = np.arange(num_out)[:, None]
    m = np.arange(n)[None, :]
    basis = 2.0 * np.cos(np.pi * k * (2 * m + 1) / (2.0 * n))
    scale = np.full((num_out, 1), math.sqrt(1.0 / (2.0 * n)))
    scale[0, 0] = math.sqrt(1.0 / (4.0 * n))
    return x @ (basis * scale).T


def lifter(cepstra, lift=CEP_LIFTER):
    n = np.arange(cepstra.shape[1])
    return cepstra * (1.0 + (lift / 2.0) * np.sin(np.pi * n / lift))


def delta(feat, span=2):
    """psf.delta: regression over +-span frames on an edge-padded copy, divisor 2*sum(i^2)."""
    denominator = 2 * sum(i * i for i in range(1, span + 1))
    padded = np.pad(feat, ((span, span), (0, 0)), mode='edge')
    out = np.zeros_like(feat)
    count = feat.shape[0]
    for offset in range(-span, span + 1):
        out += offset * padded[span + offset: span + offset + count]
    return out / denominator


def mfcc_with_delta(signal, sampling_rate=16000, num_features=NUM_FEATURES):
    """``__mfcc``: 40 cepstra (c0 replaced by log frame energy, lifter 22) + their deltas."""
    if num_features % 2 != 0:
        raise ValueError('num_features is not a multiple of 2.')
    feat, energy = filterbank_energies(signal, sampling_rate, num_features, N_FFT, F_MIN,
                                       sampling_rate / 2.0)
    cepstra = lifter(dct2_ortho(np.log(feat), num_features // 2))
    cepstra[:, 0] = np.log(energy)
    return np.concatenate([cepstra, delta(cepstra, 2)], axis=1)


def normalize(features, method):
    """``__feature_normalization``: population std, no epsilon (constant columns give NaN/inf)."""
    if method == 'none':
        return features
    if method == 'local':
        return (features - np.mean(features, axis=0)) / np.std(features, axis=0)
    if method == 'local_scalar':
        return (features - np.mean(features)) / np.std(features)
    raise ValueError('Invalid normalization method.')


def load_sample_from_pcm(audio, sampling_rate=16000, feature_type='mel',
                         feature_normalization='local', drop_every_second_frame=False):
    """Everything ``load_sample`` does after ``wavfile.read``: features in float64, cast to
    float32, optional frame drop, length taken *before* normalisation, normalisation in float32.
    Returns (f32[T, 80], int32 scalar)."""
    audio = np.asarray(audio)
    if len(audio) < 401:
        raise RuntimeError('Sample length {:,d} to short.'.format(len(audio)))
    if feature_type == 'mfcc':
        sample = mfcc_with_delta(audio, sampling_rate)
    elif feature_type == 'mel':
        sample = log_mel(audio, sampling_rate)
    else:
        raise ValueError('Unsupported feature type')
    sample = sample.astype(np.float32)
    if drop_every_second_frame:
        sample = sample[::2, :]
    sample_len = np.array(sample.shape[0], dtype=np.int32)
    return normalize(sample, feature_normalization), sample_len
