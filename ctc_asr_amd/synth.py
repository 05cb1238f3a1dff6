"""Synthetic corpus in the reference's on-disk layout, and synthetic in-memory batches.

Layout (``README.md:103-145`` of the reference, ``asr/params.py:141-155``):
``<corpus_dir>/<relative path>.wav`` 16 kHz mono int16 WAVs plus ``train.csv`` / ``dev.csv`` /
``test.csv`` with header ``path;label;length`` (label ``[a-z ]``, length in seconds).
Recipe of SURVEY.md 8d: Gaussian noise scaled to about +-3000 (non-zero variance in every mel
band so that 'local' normalisation is finite), labels uniform over ids 1..27 at 15 chars/s,
rows sorted by length, one sacrificial last row (the reference's input generator drops the
last CSV row, ``asr/input_functions.py:134``).
"""

import os

import numpy as np
from scipy.io import wavfile

from ctc_asr_amd.labels import ALPHABET
from ctc_asr_amd.params import CSV_DELIMITER, CSV_FIELDNAMES


def random_label(rng, num_chars):
    """Random transcription over ``' a..z'`` without leading/trailing/double spaces."""
    chars = []
    for i in range(num_chars):
        allowed_space = 0 < i < num_chars - 1 and chars[-1] != ' '
        ids = rng.integers(1 if allowed_space else 2, 28)
        chars.append(ALPHABET[int(ids) - 1])
    return ''.join(chars)


def random_pcm(rng, num_samples, scale=3000.0):
    return np.clip(rng.normal(size=num_samples) * scale, -32768, 32767).astype(np.int16)


def write_corpus(corpus_dir, csv_path, durations, seed=1234, sampling_rate=16000,
                 chars_per_second=15.0, subdir='synth', sacrificial_row=True):
    """Write one WAV per duration (seconds) and the CSV manifest; returns the CSV rows."""
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(corpus_dir, subdir), exist_ok=True)
    rows = []
    for idx, seconds in enumerate(sorted(durations)):
        num_samples = int(round(seconds * sampling_rate))
        rel = os.path.join(subdir, 'utt{:06d}.wav'.format(idx))
        wavfile.write(os.path.join(corpus_dir, rel), sampling_rate,
                      random_pcm(rng, num_samples))
        label = random_label(rng, max(1, int(seconds * chars_per_second)))
        rows.append((rel, label, '{:.4f}'.format(num_samples / sampling_rate)))
    if sacrificial_row and rows:
        rows.append(rows[-1])
    with open(csv_path, 'w', encoding='utf-8') as handle:
        handle.write(CSV_DELIMITER.join(CSV_FIELDNAMES) + '\n')
        for row in rows:
            handle.write(CSV_DELIMITER.join(row) + '\n')
    return rows


def librispeech_like_durations(rng, count, low=0.7, high=17.0, drop=False):
    """Log-normal utterance lengths (median 10.5 s) within the reference's [0.7, 17] s corpus
    filter (``asr/params.py:142-143``): clipped to it, or - ``drop=True``, what a corpus filter
    does - drawn again until inside it (no pile of utterances at exactly 17 s)."""
    out = rng.lognormal(mean=2.35, sigma=0.45, size=count)
    if not drop:
        return np.clip(out, low, high)
    bad = (out < low) | (out > high)
    while bad.any():
        out[bad] = rng.lognormal(mean=2.35, sigma=0.45, size=int(bad.sum()))
        bad = (out < low) | (out > high)
    return out


def synthetic_batch(batch, seconds, seed=1234, chars_per_second=15.0, num_features=80,
                    frames=None):
    """In-memory batch shaped like the input pipeline's output for fixed-length utterances:
    (features f32[B, T, 80] ~ N(0, 1) like 'local'-normalised features, lengths i32[B],
    dense zero-padded labels i32[B, L], plaintext list)."""
    from ctc_asr_amd.labels import encode
    rng = np.random.default_rng(seed)
    if frames is None:
        num_samples = int(round(seconds * 16000))
        frames = 1 + int(np.ceil((num_samples - 400) / 160.0))
    feats = rng.normal(size=(batch, frames, num_features)).astype(np.float32)
    lengths = np.full(batch, frames, dtype=np.int32)
    texts = [random_label(rng, max(1, int(seconds * chars_per_second))) for _ in range(batch)]
    width = max(len(t) for t in texts)
    labels = np.zeros((batch, width), dtype=np.int32)
    for b, text in enumerate(texts):
        labels[b, :len(text)] = encode(text)
    return feats, lengths, labels, texts
