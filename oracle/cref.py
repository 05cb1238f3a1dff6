"""ORACLE (test infrastructure): ctypes access to ``oracle/c/ctc_ref.c`` (plain-C CTC loss and
TensorFlow-style beam search).  Built by ``build()`` below (called from
``__graft_entry__.build()``); only tests, ``smoke()`` and ``bench.py``'s cpu_baseline load it."""

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, 'c', 'ctc_ref.c')
_OUT_DIR = os.path.join(_HERE, '_build')
_LIB_PATH = os.path.join(_OUT_DIR, 'libctc_ref.so')
_lib = None


def build(force=False):
    """gcc -O2 -shared -fPIC oracle/c/ctc_ref.c -> oracle/_build/libctc_ref.so"""
    os.makedirs(_OUT_DIR, exist_ok=True)
    if not force and os.path.exists(_LIB_PATH) and \
            os.path.getmtime(_LIB_PATH) >= os.path.getmtime(_SRC):
        return _LIB_PATH
    subprocess.check_call(['gcc', '-O2', '-shared', '-fPIC', '-o', _LIB_PATH, _SRC, '-lm'])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _ptr(arr, ctype):
    return arr.ctypes.data_as(ctypes.POINTER(ctype))


def pack_labels(labels):
    offsets = np.zeros(len(labels) + 1, dtype=np.int32)
    offsets[1:] = np.cumsum([len(row) for row in labels])
    flat = np.array([v for row in labels for v in row], dtype=np.int32)
    if flat.size == 0:
        flat = np.zeros(1, dtype=np.int32)
    return flat, offsets


def ctc_loss(logits, labels, seq_len, blank=None):
    """(loss f64[B], grad f64[T,B,C], status i32[B]) — see ``oracle_ctc_loss``."""
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    num_steps, batch, num_classes = logits.shape
    blank = num_classes - 1 if blank is None else blank
    flat, offsets = pack_labels(labels)
    seq_len = np.ascontiguousarray(seq_len, dtype=np.int32)
    loss = np.zeros(batch, dtype=np.float64)
    grad = np.zeros(logits.shape, dtype=np.float64)
    status = np.zeros(batch, dtype=np.int32)
    rc = lib().oracle_ctc_loss(_ptr(logits, ctypes.c_float), num_steps, batch, num_classes,
                               _ptr(flat, ctypes.c_int), _ptr(offsets, ctypes.c_int),
                               _ptr(seq_len, ctypes.c_int), blank, _ptr(loss, ctypes.c_double),
                               _ptr(grad, ctypes.c_double), _ptr(status, ctypes.c_int))
    if rc != 0:
        raise RuntimeError('oracle_ctc_loss failed')
    return loss, grad, status


def beam_search_decode(logits, seq_len, beam_width, blank=None, normalization='max'):
    """(list of B label lists, logp f32[B]) — see ``oracle_ctc_beam_decode``."""
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    num_steps, batch, num_classes = logits.shape
    blank = num_classes - 1 if blank is None else blank
    seq_len = np.ascontiguousarray(seq_len, dtype=np.int32)
    out = np.zeros((batch, max(num_steps, 1)), dtype=np.int32)
    out_len = np.zeros(batch, dtype=np.int32)
    logp = np.zeros(batch, dtype=np.float32)
    mode = {'max': 0, 'log_softmax': 1}[normalization]
    rc = lib().oracle_ctc_beam_decode(_ptr(logits, ctypes.c_float), num_steps, batch, num_classes,
                                      _ptr(seq_len, ctypes.c_int), int(beam_width), blank, mode,
                                      _ptr(out, ctypes.c_int), _ptr(out_len, ctypes.c_int),
                                      _ptr(logp, ctypes.c_float))
    if rc != 0:
        raise RuntimeError('oracle_ctc_beam_decode failed')
    return [out[b, :out_len[b]].tolist() for b in range(batch)], logp
