"""Corpus loading, feature extraction and batching.

Drop-in counterpart of ``asr/input_functions.py`` (``input_fn_generator`` :21-122,
``__input_generator`` :125-153, ``load_sample`` :156-250).  The reference computes features with
python_speech_features on one Python thread and batches with ``tf.data``; here WAV bytes are
read on the host, raw int16 PCM of a whole batch is uploaded once, and log-mel / MFCC features,
the optional frame drop and the per-utterance normalisation run on the MI355X
(``ctcasr_features``), writing straight into the zero-padded ``[B, T, 80]`` batch tensor the
model consumes.  Bucketing only needs frame *counts*, which follow from the sample count in the
WAV header: the shuffle buffer holds paths, not audio (16 384 utterances of LibriSpeech would be
~6 GB of PCM), and in data-parallel runs a rank reads only the utterances of its own shard.

Reference quirks kept on purpose: the CSV slice ``[1:-1]`` drops the
header and the LAST example; ``train_batch`` keeps CSV order and drops the remainder; bucketed
targets shuffle, pad to the longest utterance of the batch and keep partial final batches;
int16 PCM is not rescaled; padding (0.0) is appended after normalisation.
"""

import bisect
import os
import queue
import random
import threading
import wave
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch
from scipy.io import wavfile

from ctc_asr_amd import hip
from ctc_asr_amd.csv_helper import get_bucket_boundaries, read_csv_rows
from ctc_asr_amd.labels import ctoi
from ctc_asr_amd.params import CSV_HEADER_LABEL, CSV_HEADER_PATH, FLAGS

READER_THREADS = 8           # WAV files of one batch are read concurrently
SUPPORTED_FEATURE_TYPES = ('mel', 'mfcc')
SUPPORTED_NORMALIZATIONS = ('none', 'local', 'local_scalar')


def read_wav(file_path):
    """``wavfile.read`` + the reference's validity checks (``asr/input_functions.py:205-217``)."""
    if not isinstance(file_path, str):
        file_path = str(file_path, 'utf-8')
    if not os.path.isfile(file_path):
        raise ValueError('"{}" does not exist.'.format(file_path))
    sampling_rate, audio = wavfile.read(file_path)
    if len(audio) < 401:
        raise RuntimeError('Sample length {:,d} to short: {}'.format(len(audio), file_path))
    if sampling_rate != FLAGS.sampling_rate:
        raise RuntimeError('Sampling rate is {:,d}, expected {:,d}.'
                           .format(sampling_rate, FLAGS.sampling_rate))
    if audio.dtype != np.int16 or audio.ndim != 1:
        raise RuntimeError('Only mono 16-bit PCM WAV files are supported: {}'.format(file_path))
    return audio


def probe_wav(file_path):
    """Sample count of a WAV file from its header alone, with `read_wav`'s checks.  Bucketing and
    shuffling only need the length, so the samples themselves are read later - after the shuffle
    buffer and, in data-parallel runs, only by the rank that owns the utterance."""
    if not os.path.isfile(file_path):
        raise ValueError('"{}" does not exist.'.format(file_path))
    try:
        with wave.open(file_path, 'rb') as handle:
            rate, count = handle.getframerate(), handle.getnframes()
            mono16 = handle.getnchannels() == 1 and handle.getsampwidth() == 2
    except (wave.Error, EOFError):         # e.g. WAVE_FORMAT_EXTENSIBLE: take the slow road
        return len(read_wav(file_path))
    if count < 401:
        raise RuntimeError('Sample length {:,d} to short: {}'.format(count, file_path))
    if rate != FLAGS.sampling_rate:
        raise RuntimeError('Sampling rate is {:,d}, expected {:,d}.'
                           .format(rate, FLAGS.sampling_rate))
    if not mono16:
        raise RuntimeError('Only mono 16-bit PCM WAV files are supported: {}'.format(file_path))
    return count


def num_frames(num_samples, drop_every_second_frame=None):
    """Feature frames of an utterance of ``num_samples`` samples (after the optional drop)."""
    drop = FLAGS.features_drop_every_second_frame if drop_every_second_frame is None \
        else drop_every_second_frame
    frames = hip.features_num_frames(num_samples)
    return (frames + 1) // 2 if drop else frames


def _check_feature_args(feature_type, feature_normalization):
    feature_type = feature_type if feature_type is not None else FLAGS.feature_type
    feature_normalization = feature_normalization if feature_normalization is not None \
        else FLAGS.feature_normalization
    if feature_type not in SUPPORTED_FEATURE_TYPES:
        raise ValueError('Requested feature type of {} isn\'t supported.'.format(feature_type))
    if feature_normalization not in SUPPORTED_NORMALIZATIONS:
        raise ValueError('Requested feature normalization method {} is invalid.'
                         .format(feature_normalization))
    return feature_type, feature_normalization


def features_from_pcm(pcm_list, device='cuda', feature_type=None, feature_normalization=None):
    """List of int16 arrays -> (f32[B, Tmax, 80] device tensor, i32[B] device tensor)."""
    feature_type, feature_normalization = _check_feature_args(feature_type,
                                                              feature_normalization)
    lengths = np.array([len(p) for p in pcm_list], dtype=np.int32)
    batch = np.zeros((len(pcm_list), int(lengths.max())), dtype=np.int16)
    for row, pcm in enumerate(pcm_list):
        batch[row, :len(pcm)] = pcm
    return hip.features(torch.from_numpy(batch).to(device), torch.from_numpy(lengths).to(device),
                        feature_type, feature_normalization,
                        FLAGS.features_drop_every_second_frame, FLAGS.sampling_rate)


_RINGS = {}      # the live staging ring per (device, depth, batch size)


class _StagingRing:
    """Pinned host buffers of the reader thread, allocated ONCE per epoch iterator and reused
    round-robin: allocating (or freeing) pinned memory synchronises with the device - a
    `pin_memory()` per batch stalls the training steps in flight (measured: 16 ms per batch of the
    input pipeline alone, 35 instead of 17 ms per C5 step).  A slot is reused only after the
    upload that read it has finished (its event)."""

    def __init__(self, slots, batch_rows):
        self.slots = [{'pcm': None, 'small': torch.empty(1 << 16, dtype=torch.int32).pin_memory(),
                       'event': None} for _ in range(slots)]
        self.rows, self.next, self.busy = batch_rows, 0, False

    def take(self, samples):
        slot = self.slots[self.next]
        self.next = (self.next + 1) % len(self.slots)
        if slot['event'] is not None:
            slot['event'].synchronize()
        if slot['pcm'] is None or slot['pcm'].numel() < self.rows * samples:
            # (grows to the longest batch seen: a handful of allocations per epoch, not one per
            # batch; utterances are capped by the corpus filter - 17 s = 272 000 samples)
            slot['pcm'] = torch.empty(self.rows * max(samples, 1 << 16), dtype=torch.int16) \
                .pin_memory()
        return slot


class _Staged:
    """One batch on its way to HBM: zero-padded int16 PCM, sample counts and the CTC labels in
    their packed form, copied from PINNED host buffers with asynchronous copies on a stream of
    their own.  A pageable `.to(device)` is a synchronous copy in stream order: it would stall the
    host until the GPU has worked off everything enqueued before it - the previous training
    steps - and with that the run-ahead that keeps the GPU fed (train.py from disk ran 35 % under
    the same steps from HBM).  Built by the reader thread; the consumer makes
    its stream wait for ``ready`` and never blocks."""

    def __init__(self, items, device, stream, ring=None):
        pcm = [it[0] for it in items]
        lengths = np.array([len(p) for p in pcm], dtype=np.int32)
        width = int(lengths.max())
        rows = [list(it[1]) for it in items]
        offsets = np.zeros(len(rows) + 1, dtype=np.int32)
        offsets[1:] = np.cumsum([len(r) for r in rows])
        flat = np.array([v for r in rows for v in r] or [0], dtype=np.int32)
        if ring is None:
            ring = _StagingRing(1, len(pcm))
        slot = ring.take(width) if len(pcm) <= ring.rows else _StagingRing(1, len(pcm)).take(width)
        host_pcm = slot['pcm'][:len(pcm) * width].view(len(pcm), width)
        view = host_pcm.numpy()
        for row, samples in enumerate(pcm):
            view[row, :len(samples)] = samples
            view[row, len(samples):] = 0
        small = slot['small']
        need = len(lengths) + len(flat) + len(offsets)
        if small.numel() < need:
            small = slot['small'] = torch.empty(2 * need, dtype=torch.int32).pin_memory()
        cuts = np.cumsum([0, len(lengths), len(flat), len(offsets)])
        packed = small.numpy()
        for lo, hi, part in zip(cuts[:-1], cuts[1:], (lengths, flat, offsets)):
            packed[lo:hi] = part
        with torch.cuda.stream(stream):
            self.pcm = host_pcm.to(device, non_blocking=True)
            small_d = small[:need].to(device, non_blocking=True)
            self.ready = torch.cuda.Event()
            self.ready.record(stream)
        slot['event'] = self.ready
        self.num_samples = small_d[cuts[0]:cuts[1]]
        self.packed_labels = (small_d[cuts[1]:cuts[2]], small_d[cuts[2]:cuts[3]],
                              max([len(r) for r in rows] + [1]), rows)
        self._keep = small_d
        label_width = max(len(r) for r in rows)
        self.labels = np.zeros((len(rows), max(label_width, 1)), dtype=np.int32)
        for row, ids in enumerate(rows):
            self.labels[row, :len(ids)] = ids
        self.texts = [it[2] for it in items]
        self.seconds = float(lengths.sum()) / FLAGS.sampling_rate


def load_sample(file_path, feature_type=None, feature_normalization=None, device='cuda'):
    """Load a WAV file and convert it into feature vectors: (f32[T, 80] ndarray, int32 scalar).

    Same contract and error behaviour as the reference's ``load_sample``; the arithmetic runs on
    the GPU."""
    feature_type, feature_normalization = _check_feature_args(feature_type,
                                                              feature_normalization)
    audio = read_wav(file_path)
    feats, lengths = features_from_pcm([audio], device, feature_type, feature_normalization)
    sample_len = np.array(int(lengths[0]), dtype=np.int32)
    return feats[0, :int(sample_len)].cpu().numpy(), sample_len


def read_manifest(csv_path):
    """CSV rows as the reference's generator sees them: header AND last row dropped."""
    return read_csv_rows(csv_path)[1:-1]


class Batch:
    """One minibatch: ``features`` dict + dense zero-padded ``labels`` like the reference's
    ``input_fn`` output (``asr/input_functions.py:112-120``), tensors already in HBM.
    ``packed_labels``: the same labels as `CTCModel.pack_labels` would upload them (what
    `Trainer.train_step` / `loss_fn` take without another host-to-device copy); ``pcm`` /
    ``num_samples``: the raw audio the features were computed from (device tensors)."""

    def __init__(self, spectrogram, spectrogram_length, label_plaintext, labels, seconds,
                 packed_labels=None, pcm=None, num_samples=None):
        self.features = {'spectrogram': spectrogram, 'spectrogram_length': spectrogram_length,
                         'label_plaintext': label_plaintext}
        self.labels = labels
        self.audio_seconds = seconds
        self.packed_labels = packed_labels if packed_labels is not None else labels
        self.pcm, self.num_samples = pcm, num_samples

    def __iter__(self):      # (features, labels) = batch
        return iter((self.features, self.labels))


def _make_batch(items, device, staged=None):
    """Features of one batch on the current stream.  ``staged``: the batch's `_Staged` uploads
    (made by the reader thread); without it they are made here."""
    if staged is None:
        staged = _Staged(items, device, torch.cuda.current_stream(device))
    main = torch.cuda.current_stream(device)
    main.wait_event(staged.ready)
    for tensor in (staged.pcm, staged._keep):
        tensor.record_stream(main)          # allocated on the upload stream, used on this one
    feature_type, feature_normalization = _check_feature_args(None, None)
    feats, lengths = hip.features(staged.pcm, staged.num_samples, feature_type,
                                  feature_normalization, FLAGS.features_drop_every_second_frame,
                                  FLAGS.sampling_rate)
    return Batch(feats, lengths, staged.texts, staged.labels, staged.seconds,
                 staged.packed_labels, staged.pcm, staged.num_samples)


def _example_stream(csv_path, shuffle, rng):
    lines = read_manifest(csv_path)
    if shuffle:
        rng.shuffle(lines)
    for line in lines:
        path = os.path.join(FLAGS.corpus_dir, line[CSV_HEADER_PATH])
        label = line[CSV_HEADER_LABEL]
        yield path, [ctoi(c) for c in label], label, num_frames(probe_wav(path))


def _shuffle_buffer(stream, size, rng):
    """``tf.data.Dataset.shuffle(size)``: uniform draws from a sliding buffer."""
    buf = []
    for item in stream:
        if len(buf) < size:
            buf.append(item)
            continue
        idx = rng.randrange(size)
        yield buf[idx]
        buf[idx] = item
    rng.shuffle(buf)
    for item in buf:
        yield item


def _group_batches(stream, use_buckets, boundaries, batch_size):
    if not use_buckets:
        pending = []
        for item in stream:
            pending.append(item)
            if len(pending) == batch_size:
                yield pending
                pending = []
        return                                   # drop_remainder=True
    buckets = {}
    for item in stream:
        key = bisect.bisect_right(boundaries, item[3])
        bucket = buckets.setdefault(key, [])
        bucket.append(item)
        if len(bucket) == batch_size:
            yield bucket
            buckets[key] = []
    for key in sorted(buckets):                  # partial final batches are kept
        if buckets[key]:
            yield buckets[key]


def host_batches(csv_path, use_buckets, boundaries, rank=0, world_size=1, seed=None):
    """Host side of one epoch for one rank: yields lists of (int16 PCM, label ids, label text).
    Groups of ``world_size * batch_size`` utterances are formed exactly as a single process would
    form them (every rank walks the same seeded order), a rank keeps its contiguous share and only
    reads the WAV files of that share."""
    rng = random.Random(seed) if (seed is not None or world_size > 1) else random.Random()
    if world_size > 1 and seed is None:
        rng = random.Random(FLAGS.random_seed or 1)
    stream = _example_stream(csv_path, use_buckets, rng)
    if use_buckets:
        stream = _shuffle_buffer(stream, FLAGS.shuffle_buffer_size, rng)
    groups = _group_batches(stream, use_buckets, boundaries, FLAGS.batch_size * world_size)
    readers = ThreadPoolExecutor(max_workers=READER_THREADS)
    try:
        for group in groups:
            part = group
            if world_size > 1:
                per_rank = len(group) // world_size
                part = group[rank * per_rank:(rank + 1) * per_rank]
            if part:
                audio = list(readers.map(read_wav, [it[0] for it in part]))
                yield [(pcm, it[1], it[2]) for pcm, it in zip(audio, part)]
    finally:
        readers.shutdown(wait=False)


def input_fn_generator(target, device='cuda', rank=0, world_size=1, seed=None, prefetch=8):
    """Zero-argument ``input_fn`` for ``target`` in {'train_bucket', 'train_batch', 'dev', 'test'}
    (``asr/input_functions.py:21-56``).  Calling it returns an iterator of `Batch`.

    Data-parallel runs group ``world_size * batch_size`` utterances per step from ONE bucket (all
    ranks walk the same seeded order) and every rank keeps its contiguous shard, so the time
    extent matches across ranks and nobody waits in the all-reduce; a ragged last group is cut
    to a multiple of ``world_size``.
    """
    if target == 'train_bucket':
        csv_path, use_buckets = FLAGS.train_csv, True
    elif target == 'train_batch':
        csv_path, use_buckets = FLAGS.train_csv, False
    elif target == 'dev':
        csv_path, use_buckets = FLAGS.dev_csv, True
    elif target == 'test':
        csv_path, use_buckets = FLAGS.test_csv, True
    else:
        raise ValueError('Invalid target: "{}"'.format(target))
    boundaries = get_bucket_boundaries(csv_path, FLAGS.num_buckets) if use_buckets else []

    def input_fn():
        assert os.path.exists(csv_path) and os.path.isfile(csv_path)

        def host_side():
            return host_batches(csv_path, use_buckets, boundaries, rank, world_size, seed)

        if prefetch <= 0:
            for items in host_side():
                yield _make_batch(items, torch.device(device) if torch.device(device).index
                                  is not None else torch.device('cuda', torch.cuda.current_device()))
            return
        # WAV reading, label packing and the uploads run ahead on a host thread (the reference's
        # prefetch(64)): pinned staging buffers, asynchronous copies on an upload stream
        pending = queue.Queue(maxsize=prefetch)
        done = object()
        dev = torch.device(device)
        if dev.index is None:
            dev = torch.device('cuda', torch.cuda.current_device())
        upload = torch.cuda.Stream(dev)
        # (queue + producer + consumer; kept across epochs: pinned allocations cost up to 80 ms)
        # (one iterator at a time per ring: an iterator that starts while another one's reader
        # thread is still alive gets buffers of its own)
        key = (str(dev), prefetch + 3, FLAGS.batch_size)
        ring = _RINGS.get(key)
        if ring is None:
            _RINGS.clear()
            ring = _RINGS[key] = _StagingRing(prefetch + 3, FLAGS.batch_size)
        elif ring.busy:
            ring = _StagingRing(prefetch + 3, FLAGS.batch_size)
        ring.busy = True

        def producer():
            try:
                torch.cuda.set_device(dev)
                for items in host_side():
                    pending.put((items, _Staged(items, dev, upload, ring)))
                pending.put(done)
            except BaseException as exc:      # surface reader errors in the consumer
                pending.put(exc)
            finally:
                ring.busy = False

        threading.Thread(target=producer, daemon=True).start()
        while True:
            got = pending.get()
            if got is done:
                return
            if isinstance(got, BaseException):
                raise got
            yield _make_batch(got[0], dev, got[1])

    return input_fn
