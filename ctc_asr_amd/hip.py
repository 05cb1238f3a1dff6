"""ctypes binding of ``libctcasr.so`` (``include/ctcasr.h``) for torch tensors.

PyTorch only supplies device memory and the HIP stream here; every function below hands raw
device pointers to the C ABI.  There is no fallback: a missing library, a CPU tensor or a
non-zero status raises.
"""

import ctypes
import functools
import os

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, 'libctcasr.so')

ABI_VERSION = 7
BUILD_PROBE_WRONG_RESULTS, BUILD_NONDEFAULT_TUNING = 1, 2      # ctcasr_build_flags() bits
RNN_DEFAULT, RNN_HALF_CHIP, RNN_WHOLE_CHIP, RNN_ONE_BARRIER = 0, 1, 2, 4   # rnn_fwd/bwd `flags`
RNN_REDUCE_SCATTER = 8
RNN_F16 = 16          # persistent LSTM / GRU kernels: the recurrent products as fp16x3 (ABI v5)
RNN_XCD_SPLIT = 32    # fp16-pipe kernels: one direction per half of the XCDs
RNN_STAGGER = 64      # fp16-pipe LSTM-1024 backward, 24 / 32 rows: tiles staggered by half a step
RNN_KPAIR = 128       # ... and the K axis split over pairs of workgroups (ABI v7)
CELL_IDS = {'rnn_relu': 0, 'rnn_tanh': 1, 'lstm': 2, 'gru': 3}
CELL_GATES = {'rnn_relu': 1, 'rnn_tanh': 1, 'lstm': 4, 'gru': 3}

_c_int, _c_i64, _c_u64 = ctypes.c_int, ctypes.c_int64, ctypes.c_uint64
_c_f, _c_p, _c_sz = ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t

# name -> (restype, argtypes); doubles as the list of symbols the ABI must export
SIGNATURES = {
    'ctcasr_abi_version': (_c_int, []),
    'ctcasr_build_flags': (ctypes.c_uint, []),
    'ctcasr_error_string': (ctypes.c_char_p, [_c_int]),
    'ctcasr_set_option': (_c_int, [ctypes.c_char_p, _c_int]),
    'ctcasr_rnn_kernel_events': (_c_int, [_c_p, _c_p]),
    'ctcasr_log_softmax_fwd': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_p]),
    'ctcasr_log_softmax_bwd': (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_p]),
    'ctcasr_ctc_loss_workspace_bytes': (_c_sz, [_c_int] * 4),
    'ctcasr_ctc_loss_fwd_bwd': (_c_int, [_c_p] * 4 + [_c_int] * 5 + [_c_f] + [_c_p] * 4 +
                                [_c_sz, _c_p]),
    'ctcasr_ctc_greedy_decode': (_c_int, [_c_p, _c_p] + [_c_int] * 4 + [_c_p, _c_p, _c_p]),
    'ctcasr_ctc_beam_workspace_bytes': (_c_sz, [_c_int] * 4),
    'ctcasr_ctc_beam_decode': (_c_int, [_c_p, _c_p] + [_c_int] * 6 + [_c_p] * 4 + [_c_sz, _c_p]),
    'ctcasr_rnn_reserve_bytes': (_c_sz, [_c_int] * 4),
    'ctcasr_rnn_workspace_bytes': (_c_sz, [_c_int] * 4),
    'ctcasr_rnn_persistent_supported': (_c_int, [_c_int] * 4),
    'ctcasr_rnn_poll_error': (_c_int, [_c_p, _c_sz] + [_c_int] * 4 + [_c_p]),
    'ctcasr_rnn_resident_gate': (_c_int, [_c_p, _c_sz] + [_c_int] * 4 + [ctypes.c_uint, _c_int,
                                                                          _c_p]),
    'ctcasr_rnn_gru_drec_offset': (_c_sz, [_c_int] * 3),
    'ctcasr_rnn_fwd': (_c_int, [_c_int] + [_c_p] * 5 + [_c_int] * 3 + [_c_p] * 3 + [_c_sz, _c_p]),
    'ctcasr_rnn_bwd': (_c_int, [_c_int] + [_c_p] * 5 + [_c_int] * 3 + [_c_p] * 4 + [_c_sz, _c_p]),
    'ctcasr_rnn_fwd_steps': (_c_int, [_c_int] + [_c_p] * 5 + [_c_int] * 3 + [_c_p] * 4 +
                             [_c_sz, _c_int, _c_int, _c_int, _c_p]),
    'ctcasr_rnn_fwd_f16_supported': (_c_int, [_c_int] * 5),
    'ctcasr_rnn_bwd_steps': (_c_int, [_c_int] + [_c_p] * 5 + [_c_int] * 3 + [_c_p] * 5 +
                             [_c_sz, _c_int, _c_int, _c_int, _c_p]),
    'ctcasr_rnn_bwd_f16_supported': (_c_int, [_c_int] * 5),
    'ctcasr_bias_act_fwd': (_c_int, [_c_p, _c_p, _c_i64, _c_int, _c_f, _c_f, _c_u64, _c_p]),
    'ctcasr_bias_act_bwd': (_c_int, [_c_p] * 4 + [_c_i64, _c_int, _c_f, _c_f, _c_p]),
    'ctcasr_dropout': (_c_int, [_c_p, _c_p, _c_i64, _c_f, _c_u64, _c_p]),
    'ctcasr_colsum_accumulate': (_c_int, [_c_p, _c_p, _c_i64, _c_int, _c_p]),
    'ctcasr_conv_s12_supported': (_c_int, [_c_int, _c_int]),
    'ctcasr_conv_s12_pack_weights': (_c_int, [_c_p, _c_p, _c_int, _c_p]),
    'ctcasr_conv_s12_fwd': (_c_int, [_c_p, _c_p, _c_p, _c_p] + [_c_int] * 4 + [_c_f, _c_int, _c_p]),
    'ctcasr_conv_s12_pack16_bytes': (_c_sz, [_c_int]),
    'ctcasr_conv_s12_pack_weights16': (_c_int, [_c_p, _c_p, _c_int, _c_p]),
    'ctcasr_conv_s12_fwd16': (_c_int, [_c_p, _c_f, _c_p, _c_p, _c_p] + [_c_int] * 4 +
                              [_c_f, _c_int, _c_p]),
    'ctcasr_conv_s12_bwd_data16': (_c_int, [_c_p, _c_p, _c_p] + [_c_int] * 5 + [_c_p, _c_f, _c_p]),
    'ctcasr_conv0_pack16_bytes': (_c_sz, []),
    'ctcasr_conv0_pack_weights16': (_c_int, [_c_p, _c_p, _c_p]),
    'ctcasr_conv0_fwd16': (_c_int, [_c_p] * 4 + [_c_int, _c_int, _c_f, _c_p]),
    'ctcasr_conv0_wrw16_workspace_bytes': (_c_sz, [_c_int, _c_int]),
    'ctcasr_conv0_wrw16': (_c_int, [_c_p] * 3 + [_c_int, _c_int, _c_p, _c_f, _c_p, _c_p, _c_sz,
                                                 _c_p]),
    'ctcasr_conv_s12_wrw16_workspace_bytes': (_c_sz, [_c_int] * 4),
    'ctcasr_conv_s12_wrw16': (_c_int, [_c_p, _c_p, _c_f, _c_p] + [_c_int] * 5 +
                              [_c_p, _c_f, _c_p, _c_p, _c_sz, _c_p]),
    'ctcasr_conv_s12_bwd_data': (_c_int, [_c_p, _c_p, _c_p] + [_c_int] * 5 + [_c_p, _c_f, _c_p]),
    'ctcasr_conv_s12_wrw_workspace_bytes': (_c_sz, [_c_int] * 4),
    'ctcasr_conv_s12_wrw': (_c_int, [_c_p] * 3 + [_c_int] * 5 + [_c_p, _c_f, _c_p] +
                            [_c_p, _c_sz, _c_p]),
    'ctcasr_conv0_fwd': (_c_int, [_c_p, _c_p, _c_p, _c_p, _c_int, _c_int, _c_f, _c_p]),
    'ctcasr_conv0_wrw_workspace_bytes': (_c_sz, [_c_int, _c_int]),
    'ctcasr_conv0_wrw': (_c_int, [_c_p, _c_p, _c_p, _c_int, _c_int, _c_p, _c_f, _c_p, _c_p, _c_sz,
                                  _c_p]),
    'ctcasr_transpose_batched': (_c_int, [_c_p, _c_p, _c_int, _c_int, _c_int, _c_p]),
    'ctcasr_split_bf16': (_c_int, [_c_p, _c_i64, _c_int, _c_i64, _c_p, _c_int, _c_p, _c_i64, _c_i64,
                                   _c_p]),
    'ctcasr_split_f16': (_c_int, [_c_p, _c_i64, _c_int, _c_i64, _c_f, _c_p, _c_int, _c_p, _c_i64,
                                  _c_i64, _c_p]),
    'ctcasr_colmax_scale': (_c_int, [_c_p, _c_i64, _c_int, _c_i64, _c_p, _c_p, _c_p, _c_p]),
    'ctcasr_colscale_from_max': (_c_int, [_c_p, _c_int, _c_p, _c_p, _c_p]),
    'ctcasr_split_f16_cols': (_c_int, [_c_p, _c_i64, _c_int, _c_i64, _c_p, _c_f, _c_p, _c_int, _c_p,
                                       _c_i64, _c_i64, _c_p]),
    'ctcasr_split_f16_rows': (_c_int, [_c_p, _c_i64, _c_int, _c_i64, _c_p, _c_int, _c_p, _c_i64,
                                       _c_i64, _c_p, _c_p]),
    'ctcasr_rescale_rows': (_c_int, [_c_p, _c_i64, _c_p, _c_f, _c_p, _c_i64, _c_i64, _c_int, _c_int,
                                     _c_p]),
    'ctcasr_gemm_split_nt': (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_i64, _c_int, _c_int, _c_int,
                                      _c_int, _c_p]),
    'ctcasr_gemm_split_tn': (_c_int, [_c_p, _c_i64, _c_p, _c_i64, _c_p, _c_i64, _c_int, _c_int, _c_int,
                                      _c_int, _c_p]),
    'ctcasr_rnn_f16_recurrence': (_c_int, [_c_int] * 7),
    'ctcasr_collective_traffic': (_c_int, [_c_int, _c_int, _c_p, _c_p, _c_i64, _c_i64, _c_p]),
    'ctcasr_wgrad16_packed_bytes': (_c_sz, [_c_int, _c_int]),
    'ctcasr_wgrad16_pack': (_c_int, [_c_p, _c_i64, _c_i64, _c_int, _c_i64, _c_int, _c_p, _c_f, _c_p,
                                     _c_p]),
    'ctcasr_wgrad16_gemm': (_c_int, [_c_p, _c_int, _c_int, _c_p, _c_p, _c_int, _c_int, _c_f, _c_p,
                                     _c_i64, _c_p, _c_int, _c_int, _c_f, _c_p, _c_i64, _c_int, _c_p,
                                     _c_p]),
    'ctcasr_wgrad16_sync_ints': (_c_sz, [_c_int, _c_int, _c_int]),
    'ctcasr_dgrad16_packed_bytes': (_c_sz, [_c_int]),
    'ctcasr_dgrad16_pack_weights': (_c_int, [_c_p, _c_i64, _c_int, _c_int, _c_f, _c_p, _c_p]),
    'ctcasr_dgrad16_supported': (_c_int, [_c_int] * 4),
    'ctcasr_dgrad16_published_offsets': (_c_int, [_c_int, _c_int, _c_int, _c_p, _c_p]),
    'ctcasr_dgrad16_blockscaled': (_c_int, [_c_p, _c_int, _c_int, _c_int, _c_p, _c_f, _c_int, _c_p,
                                            _c_i64] + [_c_int] * 5 + [_c_p]),
    'ctcasr_features_num_frames': (_c_int, [_c_int]),
    'ctcasr_features_tables_bytes': (_c_sz, []),
    'ctcasr_features_init_tables': (_c_int, [_c_p, _c_int, _c_p]),
    'ctcasr_features_workspace_bytes': (_c_sz, [_c_int, _c_int]),
    'ctcasr_features': (_c_int, [_c_p, _c_p] + [_c_int] * 5 + [_c_p, _c_p, _c_int, _c_p, _c_p,
                                 _c_sz, _c_p]),
    'ctcasr_adam_step': (_c_int, [_c_p] * 4 + [_c_i64] + [_c_f] * 4 + [_c_i64, _c_f, _c_p, _c_p]),
    'ctcasr_step_guard': (_c_int, [_c_p, _c_p, _c_int, _c_p, _c_p, _c_p, _c_p, _c_p]),
    'ctcasr_rnn_timeout_word_offset': (_c_sz, [_c_int] * 5),
    'ctcasr_absmax': (_c_int, [_c_p, _c_i64, _c_p, _c_p]),
    'ctcasr_occupy_cus': (_c_int, [_c_int, _c_int, _c_p]),
}

_lib = None

# Optional per-entry-point timing: set ``EVENTS = {}`` and every wrapped call listed in
# ``TIMED`` is bracketed by HIP events on the launch stream (``drain_events`` returns ms sums).
EVENTS = None
TIMED = ('rnn_fwd', 'rnn_bwd', 'ctc_loss_fwd_bwd', 'dgrad16')


class _Timed:
    def __init__(self, name):
        self.name = name

    def __enter__(self):
        if EVENTS is not None and self.name in TIMED:
            self.start = torch.cuda.Event(enable_timing=True)
            self.stop = torch.cuda.Event(enable_timing=True)
            self.start.record()
        return self

    def __exit__(self, *exc):
        if EVENTS is not None and self.name in TIMED:
            self.stop.record()
            EVENTS.setdefault(self.name, []).append((self.start, self.stop))
        return False


def drain_events():
    """{name: (calls, total_ms)} for the events collected so far; call after a synchronize."""
    global EVENTS
    out = {}
    for name, pairs in (EVENTS or {}).items():
        out[name] = (len(pairs), sum(a.elapsed_time(b) for a, b in pairs))
    if EVENTS is not None:
        EVENTS = {}
    return out


class CtcAsrError(RuntimeError):
    """Non-zero status from the C ABI."""


def load(path=None):
    """Load the library once; raises if it has not been built (``__graft_entry__.build()``)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get('CTCASR_LIB') or LIB_PATH      # (A/B builds: tools/build_alt.sh)
    if not os.path.exists(path):
        raise CtcAsrError('{} is missing - build it with `python -m ctc_asr_amd.build`; the '
                          'MI355X path has no CPU fallback.'.format(path))
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the ABI does not export the symbol
        fn.restype, fn.argtypes = restype, argtypes
    if lib.ctcasr_abi_version() != ABI_VERSION:
        raise CtcAsrError('libctcasr ABI version mismatch')
    # probe builds of the kernels (tools/build_alt.sh -DPRNN_PROBE_...) compute wrong results on
    # purpose: never let one stand in for the product library by accident
    if lib.ctcasr_build_flags() & BUILD_PROBE_WRONG_RESULTS and \
            os.environ.get('CTCASR_ALLOW_PROBE_BUILD') != '1':
        raise CtcAsrError('{} is a timing-probe build (ctcasr_build_flags() = {:#x}: results are '
                          'wrong on purpose); set CTCASR_ALLOW_PROBE_BUILD=1 to load it anyway.'
                          .format(path, lib.ctcasr_build_flags()))
    _lib = lib
    return lib


def _check(code, what):
    if code != 0:
        raise CtcAsrError('{} failed: {} ({})'.format(
            what, load().ctcasr_error_string(code).decode(), code))


def _dev(tensor, dtype=torch.float32, name='tensor'):
    if tensor is None:
        return None
    if not tensor.is_cuda:
        raise CtcAsrError('{} must live in HBM (got a CPU tensor); there is no CPU path.'
                          .format(name))
    if tensor.dtype != dtype:
        raise CtcAsrError('{} must be {} (got {}).'.format(name, dtype, tensor.dtype))
    if not tensor.is_contiguous():
        raise CtcAsrError('{} must be contiguous.'.format(name))
    return tensor.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _on_tensor_device(fn):
    """Run a launching wrapper with the device of its first GPU tensor argument current, so that
    the stream it launches on (`_stream`) and the memory it is given belong to the same GPU even
    when the caller's current device is another one (``CTCModel(cfg, 'cuda:1')`` from a
    process whose current device is 0)."""
    @functools.wraps(fn)
    def wrapper(*args, **kwargs):
        for value in list(args) + list(kwargs.values()):
            if torch.is_tensor(value) and value.is_cuda:
                if value.device.index != torch.cuda.current_device():
                    with torch.cuda.device(value.device):
                        return fn(*args, **kwargs)
                break
        return fn(*args, **kwargs)
    return wrapper


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# ------------------------------------------------------------------------------------------
def set_option(name, value):
    _check(load().ctcasr_set_option(name.encode(), int(value)), 'set_option')


@_on_tensor_device
def log_softmax_fwd(x, out=None):
    rows, classes = x.numel() // x.shape[-1], x.shape[-1]
    out = torch.empty_like(x) if out is None else out
    _check(load().ctcasr_log_softmax_fwd(_dev(x, name='x'), _dev(out, name='out'), rows, classes,
                                         _stream()), 'log_softmax_fwd')
    return out


@_on_tensor_device
def log_softmax_bwd(y, dy, out=None):
    rows, classes = y.numel() // y.shape[-1], y.shape[-1]
    out = torch.empty_like(y) if out is None else out
    _check(load().ctcasr_log_softmax_bwd(_dev(y, name='y'), _dev(dy, name='dy'),
                                         _dev(out, name='out'), rows, classes, _stream()),
           'log_softmax_bwd')
    return out


def ctc_loss_workspace_bytes(num_steps, batch, classes, max_label_len):
    return load().ctcasr_ctc_loss_workspace_bytes(num_steps, batch, classes, max_label_len)


@_on_tensor_device
def ctc_loss_fwd_bwd(logits, labels, label_offsets, seq_len, max_label_len, blank=None,
                     grad_scale=1.0, loss=None, grad=None, status=None, workspace=None):
    """logits f32[T,B,C]; labels/label_offsets/seq_len int32 device tensors.
    Returns (loss f32[B], grad f32[T,B,C], status i32[B])."""
    num_steps, batch, classes = logits.shape
    blank = classes - 1 if blank is None else blank
    dev = logits.device
    loss = torch.empty(batch, dtype=torch.float32, device=dev) if loss is None else loss
    grad = torch.empty_like(logits) if grad is None else grad
    status = torch.empty(batch, dtype=torch.int32, device=dev) if status is None else status
    need = ctc_loss_workspace_bytes(num_steps, batch, classes, max_label_len)
    if workspace is None:
        workspace = _workspace(need, dev)
    with _Timed('ctc_loss_fwd_bwd'):
      _check(load().ctcasr_ctc_loss_fwd_bwd(
        _dev(logits, name='logits'), _dev(labels, torch.int32, 'labels'),
        _dev(label_offsets, torch.int32, 'label_offsets'), _dev(seq_len, torch.int32, 'seq_len'),
        num_steps, batch, classes, blank, int(max_label_len), float(grad_scale),
        _dev(loss, name='loss'), _dev(grad, name='grad'), _dev(status, torch.int32, 'status'),
        _dev(workspace, torch.uint8, 'workspace'), workspace.numel(), _stream()),
        'ctc_loss_fwd_bwd')
    return loss, grad, status


@_on_tensor_device
def ctc_greedy_decode(logits, seq_len, blank=None, out=None, out_len=None):
    num_steps, batch, classes = logits.shape
    blank = classes - 1 if blank is None else blank
    dev = logits.device
    out = torch.empty((batch, num_steps), dtype=torch.int32, device=dev) if out is None else out
    out_len = torch.empty(batch, dtype=torch.int32, device=dev) if out_len is None else out_len
    _check(load().ctcasr_ctc_greedy_decode(
        _dev(logits, name='logits'), _dev(seq_len, torch.int32, 'seq_len'), num_steps, batch,
        classes, blank, _dev(out, torch.int32, 'out'), _dev(out_len, torch.int32, 'out_len'),
        _stream()), 'ctc_greedy_decode')
    return out, out_len


def ctc_beam_workspace_bytes(num_steps, batch, classes, beam_width):
    return load().ctcasr_ctc_beam_workspace_bytes(num_steps, batch, classes, int(beam_width))


@_on_tensor_device
def ctc_beam_decode(logits, seq_len, beam_width, blank=None, normalization='max'):
    num_steps, batch, classes = logits.shape
    blank = classes - 1 if blank is None else blank
    dev = logits.device
    out = torch.empty((batch, num_steps), dtype=torch.int32, device=dev)
    out_len = torch.empty(batch, dtype=torch.int32, device=dev)
    logp = torch.empty(batch, dtype=torch.float32, device=dev)
    workspace = _workspace(load().ctcasr_ctc_beam_workspace_bytes(num_steps, batch, classes,
                                                                  int(beam_width)), dev)
    _check(load().ctcasr_ctc_beam_decode(
        _dev(logits, name='logits'), _dev(seq_len, torch.int32, 'seq_len'), num_steps, batch,
        classes, blank, int(beam_width), {'max': 0, 'log_softmax': 1}[normalization],
        _dev(out, torch.int32, 'out'), _dev(out_len, torch.int32, 'out_len'),
        _dev(logp, name='logp'), _dev(workspace, torch.uint8, 'workspace'), workspace.numel(),
        _stream()), 'ctc_beam_decode')
    if bool((out_len < 0).any()):
        raise CtcAsrError('ctc_beam_decode: prefix-tree pool exhausted')
    return out, out_len, logp


def rnn_reserve_bytes(cell, num_steps, batch, hidden):
    return load().ctcasr_rnn_reserve_bytes(CELL_IDS[cell], num_steps, batch, hidden)


def rnn_workspace_bytes(cell, num_steps, batch, hidden):
    return load().ctcasr_rnn_workspace_bytes(CELL_IDS[cell], num_steps, batch, hidden)


def rnn_kernel_events():
    """{'fwd': (launches, total_ms), 'bwd': ...} of the persistent recurrence kernels recorded
    since the last call (needs ``set_option('rnn_kernel_events', 1)``)."""
    launches = (ctypes.c_int * 2)()
    total_ms = (ctypes.c_double * 2)()
    _check(load().ctcasr_rnn_kernel_events(launches, total_ms), 'rnn_kernel_events')
    return {'fwd': (launches[0], total_ms[0]), 'bwd': (launches[1], total_ms[1])}


def rnn_persistent_supported(cell, num_steps, batch, hidden):
    return bool(load().ctcasr_rnn_persistent_supported(CELL_IDS[cell], num_steps, batch, hidden))


@_on_tensor_device
def rnn_poll_error(cell, workspace, num_steps, batch, hidden):
    """Synchronise and raise if any persistent recurrence launch on ``workspace`` since the last
    poll timed out at a grid barrier (sticky word; cleared by this call)."""
    _check(load().ctcasr_rnn_poll_error(_dev(workspace, torch.uint8, 'workspace'),
                                        workspace.numel(), CELL_IDS[cell], num_steps, batch,
                                        hidden, _stream()), 'rnn persistent kernel')


@_on_tensor_device
def rnn_resident_gate(cell, workspace, num_steps, batch, hidden, ticket, max_wait_us=200):
    """Make the CURRENT stream wait (bounded) until the persistent launch carrying ``ticket``
    (``rnn_fwd`` / ``rnn_bwd(..., ticket=...)``) on ``workspace`` has all its workgroups running:
    work enqueued behind the gate cannot take the CUs that launch needs."""
    _check(load().ctcasr_rnn_resident_gate(_dev(workspace, torch.uint8, 'workspace'),
                                           workspace.numel(), CELL_IDS[cell], num_steps, batch,
                                           hidden, int(ticket), int(max_wait_us), _stream()),
           'rnn_resident_gate')


def rnn_gru_drec(reserve, num_steps, batch, hidden):
    """View of drec f32[T, B, 2, 3H] inside a GRU reserve (valid after `rnn_bwd`)."""
    start = load().ctcasr_rnn_gru_drec_offset(num_steps, batch, hidden)
    count = num_steps * batch * 2 * 3 * hidden
    return reserve[start:start + 4 * count].view(torch.float32).view(num_steps, batch, 2,
                                                                      3 * hidden)


@_on_tensor_device
def rnn_workspace(cell, num_steps, batch, hidden, device):
    """A zero-filled workspace for `rnn_fwd` / `rnn_bwd` (the sticky time-out word of the
    persistent kernels must start at zero; launches never clear it, `rnn_poll_error` does)."""
    return torch.zeros(max(int(rnn_workspace_bytes(cell, num_steps, batch, hidden)), 256),
                       dtype=torch.uint8, device=device)


@_on_tensor_device
def rnn_fwd(cell, xw, w_hh, seq_len=None, b_hh_n=None, y=None, reserve=None, workspace=None,
            steps=None, flags=RNN_DEFAULT, xw_bias=None, ticket=0, y16=None):
    """xw f32[T,B,2,G*H], w_hh f32[2,G*H,H] -> (y f32[T,B,2H], reserve, workspace).

    ``steps=(begin, end)`` runs that range of recurrence steps only (`ctcasr_rnn_fwd_steps`): cut
    a pass into calls covering 0..T in ascending order, passing the same ``y``, ``reserve`` and
    ``workspace`` to each.  ``flags``: RNN_DEFAULT / RNN_HALF_CHIP / RNN_WHOLE_CHIP (which
    variant of the persistent kernel runs - a per-call choice, no process-wide state).
    ``xw_bias`` f32[2*G*H] (optional) is added to xw inside the kernel.  ``ticket`` (1..2^24-1):
    the persistent launch posts it once all its workgroups run (`rnn_resident_gate`)."""
    num_steps, batch = xw.shape[0], xw.shape[1]
    hidden = w_hh.shape[2]
    dev = xw.device
    begin, end = (0, num_steps) if steps is None else steps
    if (begin, end) != (0, num_steps) and (y is None or reserve is None or workspace is None):
        raise ValueError('rnn_fwd: a partial step range needs the caller\'s y, reserve and '
                         'workspace (they carry the pass from one call to the next).')
    y = torch.empty((num_steps, batch, 2 * hidden), dtype=torch.float32, device=dev) \
        if y is None else y
    if reserve is None:
        reserve = _workspace(rnn_reserve_bytes(cell, num_steps, batch, hidden), dev)
    if workspace is None:
        workspace = rnn_workspace(cell, num_steps, batch, hidden, dev)
    with _Timed('rnn_fwd'):
      _check(load().ctcasr_rnn_fwd_steps(
        CELL_IDS[cell], _dev(xw, name='xw'), _dev(xw_bias, name='xw_bias'),
        _dev(w_hh, name='w_hh'), _dev(b_hh_n, name='b_hh_n'),
        _dev(seq_len, torch.int32, 'seq_len'), num_steps, batch, hidden, _dev(y, name='y'),
        _dev(y16, torch.float16, 'y16'),
        _dev(reserve, torch.uint8, 'reserve'), _dev(workspace, torch.uint8, 'workspace'),
        workspace.numel(), int(begin), int(end), int(flags) | (int(ticket) & 0xFFFFFF) << 8,
        _stream()), 'rnn_fwd')
    return y, reserve, workspace


@_on_tensor_device
def rnn_bwd(cell, dy, y, w_hh_t, reserve, seq_len=None, b_hh_n=None, dxw=None, dbias=None,
            workspace=None, steps=None, flags=RNN_DEFAULT, ticket=0, colmax=None):
    """dy,y f32[T,B,2H], w_hh_t f32[2,H,G*H] -> dxw f32[T,B,2,G*H].

    ``colmax`` (optional; only where `rnn_bwd_f16_supported`): int32[2*G*H] zeroed by the caller,
    raised to the bit patterns of the largest |dxw| per column over the steps of the call.

    ``steps=(begin, end)`` runs that range of recurrence steps only (`ctcasr_rnn_bwd_steps`): cut
    a pass into calls covering T..0 in descending order, passing the same ``dxw`` and
    ``workspace`` to each.  ``dbias`` (optional, zeroed by the caller before the pass): the bias
    gradients are accumulated into it - f32[2*G*H] column sums of dxw, then for the GRU f32[2*3H]
    column sums of drec - complete after the call that covers step 0."""
    num_steps, batch = dy.shape[0], dy.shape[1]
    hidden = w_hh_t.shape[1]
    gates = CELL_GATES[cell]
    dev = dy.device
    begin, end = (0, num_steps) if steps is None else steps
    if (begin, end) != (0, num_steps) and (dxw is None or workspace is None):
        raise ValueError('rnn_bwd: a partial step range needs the caller\'s dxw and workspace '
                         '(they carry the pass from one call to the next).')
    dxw = torch.empty((num_steps, batch, 2, gates * hidden), dtype=torch.float32, device=dev) \
        if dxw is None else dxw
    if workspace is None:
        workspace = rnn_workspace(cell, num_steps, batch, hidden, dev)
    with _Timed('rnn_bwd'):
      _check(load().ctcasr_rnn_bwd_steps(
        CELL_IDS[cell], _dev(dy, name='dy'), _dev(y, name='y'), _dev(w_hh_t, name='w_hh_t'),
        _dev(b_hh_n, name='b_hh_n'), _dev(seq_len, torch.int32, 'seq_len'), num_steps, batch,
        hidden, _dev(reserve, torch.uint8, 'reserve'), _dev(dxw, name='dxw'),
        _dev(dbias, name='dbias'), _dev(colmax, torch.int32, 'colmax'),
        _dev(workspace, torch.uint8, 'workspace'),
        workspace.numel(), int(begin), int(end), int(flags) | (int(ticket) & 0xFFFFFF) << 8,
        _stream()), 'rnn_bwd')
    return dxw


def rnn_fwd_f16_supported(cell, num_steps, batch, hidden, flags=RNN_F16):
    """Whether `rnn_fwd` with these flags runs the fp16-pipe kernel (and can write ``y16``: fp16
    [T*B, 3, 2H], the pieces of y * 2^15 in the layout of `split_f16(order (0, 0, 1))`)."""
    return bool(load().ctcasr_rnn_fwd_f16_supported(CELL_IDS[cell], int(num_steps), int(batch),
                                                    int(hidden), int(flags)))


def rnn_f16_recurrence(cell, num_steps, batch, hidden, flags=RNN_F16, backward=False,
                       ragged=False):
    """Whether a recurrence call with these flags runs an fp16-pipe kernel (include/ctcasr.h)."""
    return bool(load().ctcasr_rnn_f16_recurrence(CELL_IDS[cell], int(num_steps), int(batch),
                                                 int(hidden), int(flags), int(backward),
                                                 int(ragged)))


def rnn_bwd_f16_supported(cell, num_steps, batch, hidden, flags=RNN_F16):
    """Whether `rnn_bwd` with these flags runs the fp16-pipe kernel (and can fill ``colmax``)."""
    return bool(load().ctcasr_rnn_bwd_f16_supported(CELL_IDS[cell], int(num_steps), int(batch),
                                                    int(hidden), int(flags)))


@_on_tensor_device
def bias_act_fwd(y, bias, cutoff, dropout_rate=0.0, seed=0):
    """In place: y = dropout(min(max(y + bias, 0), cutoff)); cutoff <= 0 -> bias add only."""
    cols = y.shape[-1]
    _check(load().ctcasr_bias_act_fwd(_dev(y, name='y'), _dev(bias, name='bias'),
                                      y.numel() // cols, cols, float(cutoff), float(dropout_rate),
                                      int(seed) & 0xFFFFFFFFFFFFFFFF, _stream()), 'bias_act_fwd')
    return y


@_on_tensor_device
def bias_act_bwd(y, dy, cutoff, dropout_rate=0.0, dbias=None, dz=None):
    cols = y.shape[-1]
    dz = torch.empty_like(dy) if dz is None else dz
    _check(load().ctcasr_bias_act_bwd(_dev(y, name='y'), _dev(dy, name='dy'), _dev(dz, name='dz'),
                                      _dev(dbias, name='dbias'), y.numel() // cols, cols,
                                      float(cutoff), float(dropout_rate), _stream()),
           'bias_act_bwd')
    return dz


@_on_tensor_device
def dropout(src, rate, seed, out=None):
    """out = src * mask(seed) / (1 - rate); same call (same seed) back-propagates a gradient."""
    out = torch.empty_like(src) if out is None else out
    _check(load().ctcasr_dropout(_dev(src, name='src'), _dev(out, name='out'), src.numel(),
                                 float(rate), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream()),
           'dropout')
    return out


@_on_tensor_device
def colsum_accumulate(dz, dbias):
    cols = dz.shape[-1]
    _check(load().ctcasr_colsum_accumulate(_dev(dz, name='dz'), _dev(dbias, name='dbias'),
                                           dz.numel() // cols, cols, _stream()),
           'colsum_accumulate')
    return dbias


SPLIT_MAX_BLOCKS = 6


@_on_tensor_device
def split_bf16(x, order, out=None):
    """x f32 [rows, cols] (unit column stride, any row stride) -> bf16 ``out[rows, len(order),
    cols]``: block b holds piece ``order[b]`` of the three-piece bfloat16 split x = x1 + x2 + x3
    (include/ctcasr.h: ctcasr_split_bf16).  ``out`` may be any bf16 view of that shape with unit
    column stride (a row / column range of a bigger buffer, blocks stacked along the rows...)."""
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda or x.stride(1) != 1:
        raise CtcAsrError('split_bf16: x must be an f32 matrix in HBM with unit column stride.')
    rows, cols = x.shape
    order = [int(v) for v in order]
    if out is None:
        out = torch.empty((rows, len(order), cols), dtype=torch.bfloat16, device=x.device)
    elif (out.dtype != torch.bfloat16 or tuple(out.shape) != (rows, len(order), cols) or
          out.stride(2) != 1 or out.device != x.device):
        raise CtcAsrError('split_bf16: out must be a bf16 [rows, blocks, cols] view with unit '
                          'column stride on the device of x.')
    arr = (ctypes.c_int * len(order))(*order)
    _check(load().ctcasr_split_bf16(x.data_ptr(), rows, cols, x.stride(0) if rows > 1 else cols,
                                    arr, len(order), out.data_ptr(),
                                    out.stride(0) if rows > 1 else len(order) * cols,
                                    out.stride(1) if len(order) > 1 else cols, _stream()),
           'split_bf16')
    return out


@_on_tensor_device
def split_f16(x, scale, order, out=None):
    """x f32 [rows, cols] -> fp16 ``out[rows, len(order), cols]``: block b holds piece ``order[b]``
    (0 / 1) of the two-piece fp16 split of x * scale (include/ctcasr.h: ctcasr_split_f16).  The
    caller guarantees |x| * scale < 65504."""
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda or x.stride(1) != 1:
        raise CtcAsrError('split_f16: x must be an f32 matrix in HBM with unit column stride.')
    rows, cols = x.shape
    order = [int(v) for v in order]
    if out is None:
        out = torch.empty((rows, len(order), cols), dtype=torch.float16, device=x.device)
    elif (out.dtype != torch.float16 or tuple(out.shape) != (rows, len(order), cols) or
          out.stride(2) != 1 or out.device != x.device):
        raise CtcAsrError('split_f16: out must be an fp16 [rows, blocks, cols] view with unit '
                          'column stride on the device of x.')
    arr = (ctypes.c_int * len(order))(*order)
    _check(load().ctcasr_split_f16(x.data_ptr(), rows, cols, x.stride(0) if rows > 1 else cols,
                                   float(scale), arr, len(order), out.data_ptr(),
                                   out.stride(0) if rows > 1 else len(order) * cols,
                                   out.stride(1) if len(order) > 1 else cols, _stream()),
           'split_f16')
    return out


def _f32_matrix(t, name):
    if t.dim() != 2 or t.dtype != torch.float32 or not t.is_cuda or t.stride(1) != 1:
        raise CtcAsrError('{} must be an f32 matrix in HBM with unit column stride.'.format(name))


def _f16_out(out, rows, blocks, cols, device):
    if out is None:
        return torch.empty((rows, blocks, cols), dtype=torch.float16, device=device)
    if (out.dtype != torch.float16 or tuple(out.shape) != (rows, blocks, cols) or
            out.stride(2) != 1 or out.device != device):
        raise CtcAsrError('out must be an fp16 [rows, blocks, cols] view with unit column stride.')
    return out


@_on_tensor_device
def colscale_from_max(max_bits):
    """(scale, inv_scale) f32[cols] of `colmax_scale` from column maxima already known: int32[cols]
    bit patterns of max |x[:, c]| (what `rnn_bwd(..., colmax=)` accumulates)."""
    cols = max_bits.numel()
    scale = torch.empty(cols, dtype=torch.float32, device=max_bits.device)
    inv_scale = torch.empty(cols, dtype=torch.float32, device=max_bits.device)
    _check(load().ctcasr_colscale_from_max(_dev(max_bits, torch.int32, 'max_bits'), cols,
                                           scale.data_ptr(), inv_scale.data_ptr(), _stream()),
           'colscale_from_max')
    return scale, inv_scale


@_on_tensor_device
def colmax_scale(x, scale=None, inv_scale=None):
    """Power-of-two scale per column of x f32 [rows, cols] (largest magnitude -> [2^13, 2^14)) and
    its inverse: (scale f32[cols], inv_scale f32[cols]) (ctcasr_colmax_scale)."""
    _f32_matrix(x, 'colmax_scale: x')
    rows, cols = x.shape
    scale = torch.empty(cols, dtype=torch.float32, device=x.device) if scale is None else scale
    inv_scale = torch.empty(cols, dtype=torch.float32, device=x.device) \
        if inv_scale is None else inv_scale
    work = torch.empty(cols, dtype=torch.int32, device=x.device)
    _check(load().ctcasr_colmax_scale(x.data_ptr(), rows, cols, x.stride(0) if rows > 1 else cols,
                                      work.data_ptr(), _dev(scale, name='scale'),
                                      _dev(inv_scale, name='inv_scale'), _stream()), 'colmax_scale')
    return scale, inv_scale


@_on_tensor_device
def split_f16_cols(x, col_scale, scale, order, out=None):
    """Two fp16 pieces of x[r][c] * col_scale[c] * scale -> fp16 [rows, len(order), cols]."""
    _f32_matrix(x, 'split_f16_cols: x')
    rows, cols = x.shape
    order = [int(v) for v in order]
    out = _f16_out(out, rows, len(order), cols, x.device)
    arr = (ctypes.c_int * len(order))(*order)
    _check(load().ctcasr_split_f16_cols(
        x.data_ptr(), rows, cols, x.stride(0) if rows > 1 else cols, _dev(col_scale, name='scale'),
        float(scale), arr, len(order), out.data_ptr(),
        out.stride(0) if rows > 1 else len(order) * cols,
        out.stride(1) if len(order) > 1 else cols, _stream()), 'split_f16_cols')
    return out


@_on_tensor_device
def split_f16_rows(x, order, out=None, inv_scale=None):
    """Two fp16 pieces of every row of x scaled by that row's own power of two -> (fp16 [rows,
    len(order), cols], inv_scale f32[rows])."""
    _f32_matrix(x, 'split_f16_rows: x')
    rows, cols = x.shape
    order = [int(v) for v in order]
    out = _f16_out(out, rows, len(order), cols, x.device)
    inv_scale = torch.empty(rows, dtype=torch.float32, device=x.device) \
        if inv_scale is None else inv_scale
    arr = (ctypes.c_int * len(order))(*order)
    _check(load().ctcasr_split_f16_rows(
        x.data_ptr(), rows, cols, x.stride(0) if rows > 1 else cols, arr, len(order),
        out.data_ptr(), out.stride(0) if rows > 1 else len(order) * cols,
        out.stride(1) if len(order) > 1 else cols, _dev(inv_scale, name='inv_scale'), _stream()),
        'split_f16_rows')
    return out, inv_scale


@_on_tensor_device
def rescale_rows(t, row_factor, alpha, out, accumulate=False):
    """out[r][c] (+)= t[r][c] * row_factor[r] * alpha."""
    _f32_matrix(t, 'rescale_rows: t')
    _f32_matrix(out, 'rescale_rows: out')
    rows, cols = t.shape
    if tuple(out.shape) != (rows, cols) or row_factor.numel() != rows:
        raise CtcAsrError('rescale_rows: shapes disagree.')
    _check(load().ctcasr_rescale_rows(
        t.data_ptr(), t.stride(0) if rows > 1 else cols, _dev(row_factor, name='row_factor'),
        float(alpha), out.data_ptr(), out.stride(0) if rows > 1 else cols, rows, cols,
        int(accumulate), _stream()), 'rescale_rows')
    return out


@_on_tensor_device
def gemm_split_nt(a, b, out=None, accumulate=False):
    """out[M, N] (+)= a[M, K] . b[N, K]^T, fp32 matrices with unit column stride (row strides
    free), computed on the bf16 matrix pipe from in-register three-piece splits
    (include/ctcasr.h: ctcasr_gemm_split_nt)."""
    for name, t in (('a', a), ('b', b)):
        if t.dim() != 2 or t.dtype != torch.float32 or not t.is_cuda or t.stride(1) != 1:
            raise CtcAsrError('gemm_split_nt: {} must be an f32 matrix in HBM with unit column '
                              'stride.'.format(name))
    m, k = a.shape
    n = b.shape[0]
    if b.shape[1] != k:
        raise CtcAsrError('gemm_split_nt: a [M, K] and b [N, K] disagree on K.')
    if out is None:
        if accumulate:
            raise CtcAsrError('gemm_split_nt: accumulate needs out.')
        out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    elif (out.dtype != torch.float32 or tuple(out.shape) != (m, n) or out.stride(1) != 1 or
          out.device != a.device):
        raise CtcAsrError('gemm_split_nt: out must be an f32 [M, N] view with unit column stride.')
    _check(load().ctcasr_gemm_split_nt(a.data_ptr(), a.stride(0) if m > 1 else k, b.data_ptr(),
                                       b.stride(0) if n > 1 else k, out.data_ptr(),
                                       out.stride(0) if m > 1 else n, m, n, k, int(accumulate),
                                       _stream()), 'gemm_split_nt')
    return out


def dgrad16_supported(cell, num_steps, batch, hidden):
    """Whether the block-scaled data-gradient kernel covers a backward pass of this shape (it reads
    what the fp16-pipe LSTM-1024 backward recurrence published; include/ctcasr.h, ABI v6)."""
    return bool(load().ctcasr_dgrad16_supported(CELL_IDS[cell], int(num_steps), int(batch),
                                                int(hidden)))


@_on_tensor_device
def wgrad16_pack(x, rows_total, row0, stages, scale, col_scale=None, out=None):
    """Rows [row0, row0 + 32 * stages) of ``x`` (an f32 matrix view [>= rows_total, cols] with unit
    column stride; rows outside [0, rows_total) read as zeros), every column times col_scale[c]
    times scale, as fp16 pieces transposed into MFMA fragment order (include/ctcasr.h:
    ctcasr_wgrad16_pack) - an operand of `wgrad16_gemm`.  uint8 buffer."""
    if x.dim() != 2 or x.dtype != torch.float32 or not x.is_cuda or x.stride(1) != 1:
        raise CtcAsrError('wgrad16_pack: x must be an f32 matrix in HBM with unit column stride.')
    cols = x.shape[1]
    need = load().ctcasr_wgrad16_packed_bytes(int(stages), cols)
    if out is None:
        out = torch.empty(need, dtype=torch.uint8, device=x.device)
    elif out.dtype != torch.uint8 or out.numel() < need or not out.is_contiguous():
        raise CtcAsrError('wgrad16_pack: out must be a contiguous uint8 buffer of {} bytes.'
                          .format(need))
    _check(load().ctcasr_wgrad16_pack(
        x.data_ptr(), x.stride(0), int(rows_total), cols, int(row0), int(stages),
        _dev(col_scale, name='col_scale'), float(scale), out.data_ptr(), _stream()),
        'wgrad16_pack')
    return out


@_on_tensor_device
def wgrad16_gemm(d_packed, m, stages, inv_scale, x_packed, x_stage0, x_scale, dw_x,
                 y_packed=None, y_stage0=0, y_scale=1.0, dw_y=None, parts=1):
    """dw_x [m, nx] += D^T X and (optional) dw_y [m, ny] += D^T Y over `stages` stages of 32 rows,
    from operands packed by `wgrad16_pack` (include/ctcasr.h: ctcasr_wgrad16_gemm).  `parts`
    workgroups per tile add in order through the device's zeroed sync words (one buffer per device:
    launches of different streams / models that meet at a tile's word take turns - part 0 waits
    for the word to read 0)."""
    parts = max(1, min(int(parts), int(stages)))
    sync = None
    if parts > 1:
        need = load().ctcasr_wgrad16_sync_ints(int(m), dw_x.shape[1], 0 if dw_y is None else dw_y.shape[1])
        sync = _WGRAD16_SYNC.get(dw_x.device.index)
        if sync is None or sync.numel() < need:
            sync = _WGRAD16_SYNC[dw_x.device.index] = torch.zeros(max(need, 4096), dtype=torch.int32,
                                                            device=dw_x.device)
    for name, t in (('dw_x', dw_x), ('dw_y', dw_y)):
        if t is not None and (t.dim() != 2 or t.dtype != torch.float32 or not t.is_cuda or
                              t.stride(1) != 1 or t.shape[0] != m):
            raise CtcAsrError('wgrad16_gemm: {} must be an f32 [m, n] view in HBM with unit '
                              'column stride.'.format(name))
    _check(load().ctcasr_wgrad16_gemm(
        d_packed.data_ptr(), int(m), int(stages), _dev(inv_scale, name='inv_scale'),
        x_packed.data_ptr(), int(x_stage0), dw_x.shape[1], float(x_scale), dw_x.data_ptr(),
        dw_x.stride(0), None if y_packed is None else y_packed.data_ptr(), int(y_stage0),
        0 if dw_y is None else dw_y.shape[1], float(y_scale),
        None if dw_y is None else dw_y.data_ptr(), 0 if dw_y is None else dw_y.stride(0),
        parts, None if sync is None else sync.data_ptr(), _stream()), 'wgrad16_gemm')


_WGRAD16_SYNC = {}


def _wgrad16_sync_of(device):
    device = torch.device(device)
    return _WGRAD16_SYNC.get(torch.cuda.current_device() if device.index is None else device.index)


def wgrad16_gave_up_waiting(device):
    """True if a part of a `wgrad16_gemm` launch on `device` ever stopped waiting for its turn
    (the sticky word 0 of the sync words; synchronises)."""
    sync = _wgrad16_sync_of(device)
    return bool(sync is not None and int(sync[0].item()) != 0)


def wgrad16_check(device):
    """Raise `CtcAsrError` if a part of a `wgrad16_gemm` launch on `device` gave up waiting for
    its turn since the last check (it, and every part after it, left without adding: the weight
    gradients of that step are incomplete - `step_guard` has dropped its update), and hand the
    words back zeroed.  Synchronises; `CTCModel.check_rnn_error` calls it with the recurrence
    kernels' time-out poll."""
    sync = _wgrad16_sync_of(device)
    if sync is not None and int(sync[0].item()) != 0:
        sync.zero_()
        torch.cuda.synchronize(sync.device)
        raise CtcAsrError('wgrad16_gemm: a part of a weight-gradient tile timed out waiting for '
                          'its turn to add; the step\'s update was not applied ({}).'.format(
                              load().ctcasr_error_string(-5).decode()))



def dgrad16_packed_bytes(n):
    return int(load().ctcasr_dgrad16_packed_bytes(int(n)))


def dgrad16_published_offsets(num_steps, batch, hidden):
    """(exchange, inverse scales) byte offsets inside a recurrence workspace of what the fp16-pipe
    backward recurrence publishes for a pass over (num_steps, batch)."""
    exchange, scales = ctypes.c_size_t(), ctypes.c_size_t()
    _check(load().ctcasr_dgrad16_published_offsets(int(num_steps), int(batch), int(hidden),
                                                   ctypes.byref(exchange), ctypes.byref(scales)),
           'dgrad16_published_offsets')
    return exchange.value, scales.value


@_on_tensor_device
def dgrad16_pack_weights(w_ih, hidden, scale, out=None):
    """fp16 pieces of ``w_ih`` [2 * 4 * hidden, n] * scale in the K order / fragment order the
    block-scaled data-gradient kernel reads (uint8 buffer of ctcasr_dgrad16_packed_bytes(n))."""
    if (w_ih.dim() != 2 or w_ih.dtype != torch.float32 or not w_ih.is_cuda or
            w_ih.stride(1) != 1 or w_ih.shape[0] != 8 * hidden):
        raise CtcAsrError('dgrad16_pack_weights: w_ih must be an f32 [2 * 4H, n] matrix in HBM '
                          'with unit column stride.')
    n = w_ih.shape[1]
    need = load().ctcasr_dgrad16_packed_bytes(n)
    if out is None:
        out = torch.empty(need, dtype=torch.uint8, device=w_ih.device)
    elif out.dtype != torch.uint8 or out.numel() < need or not out.is_contiguous() or \
            out.device != w_ih.device:
        raise CtcAsrError('dgrad16_pack_weights: out must be a contiguous uint8 buffer of {} bytes.'
                          .format(need))
    _check(load().ctcasr_dgrad16_pack_weights(w_ih.data_ptr(), w_ih.stride(0), int(hidden), n,
                                              float(scale), out.data_ptr(), _stream()),
           'dgrad16_pack_weights')
    return out


@_on_tensor_device
def dgrad16_blockscaled(workspace, num_steps, batch, hidden, packed, scale, n, out=None,
                        steps=None, dirs=(0, 2), accumulate=False):
    """dx [T * B, n] (+)= dxw . W_ih from the fp16 pieces an fp16-pipe backward recurrence pass
    over (num_steps, batch) left in ``workspace`` and the packed pieces of W_ih
    (`dgrad16_pack_weights`); ``steps`` = (t_lo, t_hi) restricts the rows, ``dirs`` the directions
    whose gate columns are summed (include/ctcasr.h: ctcasr_dgrad16_blockscaled)."""
    if not workspace.is_cuda or not packed.is_cuda:
        raise CtcAsrError('dgrad16_blockscaled: workspace and packed weights must live in HBM.')
    rows = num_steps * batch
    if out is None:
        if accumulate:
            raise CtcAsrError('dgrad16_blockscaled: accumulate needs out.')
        out = torch.empty((rows, n), dtype=torch.float32, device=workspace.device)
    elif (out.dtype != torch.float32 or tuple(out.shape) != (rows, n) or out.stride(1) != 1 or
          out.device != workspace.device):
        raise CtcAsrError('dgrad16_blockscaled: out must be an f32 [T * B, n] view with unit '
                          'column stride.')
    t_lo, t_hi = (0, num_steps) if steps is None else steps
    with _Timed('dgrad16'):
        _check(load().ctcasr_dgrad16_blockscaled(
            workspace.data_ptr(), int(num_steps), int(batch), int(hidden), packed.data_ptr(),
            float(scale), int(n), out.data_ptr(), out.stride(0), int(t_lo), int(t_hi),
            int(dirs[0]), int(dirs[1]), int(accumulate), _stream()), 'dgrad16_blockscaled')
    return out


@_on_tensor_device
def gemm_split_tn(a, b, out, accumulate=True):
    """out[M, N] (+)= a[K, M]^T . b[K, N] - a product over the ROW axis of both fp32 operands
    (weight gradients) - on the bf16 matrix pipe from in-register three-piece splits
    (include/ctcasr.h: ctcasr_gemm_split_tn).  Unit column strides, any row strides, any K."""
    for name, t in (('a', a), ('b', b), ('out', out)):
        if t.dim() != 2 or t.dtype != torch.float32 or not t.is_cuda or t.stride(1) != 1:
            raise CtcAsrError('gemm_split_tn: {} must be an f32 matrix in HBM with unit column '
                              'stride.'.format(name))
    k, m = a.shape
    n = b.shape[1]
    if b.shape[0] != k or tuple(out.shape) != (m, n):
        raise CtcAsrError('gemm_split_tn: a [K, M], b [K, N] and out [M, N] disagree.')
    _check(load().ctcasr_gemm_split_tn(a.data_ptr(), a.stride(0) if k > 1 else m, b.data_ptr(),
                                       b.stride(0) if k > 1 else n, out.data_ptr(),
                                       out.stride(0) if m > 1 else n, m, n, k, int(accumulate),
                                       _stream()), 'gemm_split_tn')
    return out


def conv_s12_supported(freq_in, cout):
    """Whether the own 11x21 / stride (1,2) convolution kernels cover this layer shape."""
    return bool(load().ctcasr_conv_s12_supported(int(freq_in), int(cout)))


def conv_s12_packed_floats(cout):
    return 2 * 11 * 21 * 32 * cout


@_on_tensor_device
def conv_s12_pack_weights(weight, packed=None):
    """Fragment-ordered copies (backward, forward) of w f32[cout,32,11,21] ([Cout,Cin,kt,kf])."""
    cout = weight.shape[0]
    if tuple(weight.shape[1:]) != (32, 11, 21) or cout not in (32, 96):
        raise CtcAsrError('the conv_s12 kernels cover w [32|96, 32, 11, 21] only.')
    packed = torch.empty(conv_s12_packed_floats(cout), dtype=torch.float32,
                         device=weight.device) if packed is None else packed
    _check(load().ctcasr_conv_s12_pack_weights(_dev(weight, name='weight'),
                                               _dev(packed, name='packed'), cout, _stream()),
           'conv_s12_pack_weights')
    return packed


@_on_tensor_device
def conv_s12_fwd(x, packed, cout, bias=None, out=None, relu_cutoff=0.0, time_major=False):
    """x f32[B,T,F,32] (NHWC) -> conv(x) + bias, f32[B,T,F/2,cout]; 11x21 taps, stride (1,2),
    TensorFlow SAME padding.  ``packed`` from `conv_s12_pack_weights`.  ``relu_cutoff`` > 0 fuses
    min(max(., 0), cutoff) into the epilogue; ``time_major`` writes [T,B,F/2,cout] instead."""
    batch, frames, freq = x.shape[0], x.shape[1], x.shape[2]
    if x.shape[3] != 32 or not conv_s12_supported(freq, cout):
        raise CtcAsrError('conv_s12_fwd: unsupported layer shape {} -> {} channels'.format(
            tuple(x.shape), cout))
    shape = (frames, batch, freq // 2, cout) if time_major else (batch, frames, freq // 2, cout)
    out = torch.empty(shape, dtype=torch.float32, device=x.device) if out is None else out
    with _Timed('conv_s12_fwd'):
        _check(load().ctcasr_conv_s12_fwd(_dev(x, name='x'), _dev(packed, name='packed'),
                                          _dev(bias, name='bias'), _dev(out, name='y'), batch,
                                          frames, freq, cout, float(relu_cutoff),
                                          1 if time_major else 0, _stream()), 'conv_s12_fwd')
    return out


@_on_tensor_device
def conv_s12_pack_weights16(weight, packed=None):
    """weight f32[cout, 32, 11, 21] -> uint8 buffer for `conv_s12_fwd16`: the bit pattern of
    max |w| (found on the device) and the two fp16 pieces of w * s_w in fragment order."""
    cout = weight.shape[0]
    nbytes = load().ctcasr_conv_s12_pack16_bytes(int(cout))
    if nbytes == 0 or tuple(weight.shape[1:]) != (32, 11, 21):
        raise CtcAsrError('conv_s12_pack_weights16: unsupported kernel shape {}'.format(
            tuple(weight.shape)))
    if packed is None or packed.numel() != nbytes:
        packed = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    _check(load().ctcasr_conv_s12_pack_weights16(_dev(weight, name='weight'),
                                                 _dev(packed, torch.uint8, 'packed'), int(cout),
                                                 _stream()), 'conv_s12_pack_weights16')
    return packed


@_on_tensor_device
def conv_s12_fwd16(x, x_scale, packed16, cout, bias=None, out=None, relu_cutoff=0.0,
                   time_major=False):
    """`conv_s12_fwd` with its products on the fp16 matrix pipe (two fp16 pieces per operand,
    three products, fp32 accumulation): for x with a known bound, bound * x_scale < 65504
    (``x_scale`` a power of two).  ``packed16`` from `conv_s12_pack_weights16`."""
    batch, frames, freq = x.shape[0], x.shape[1], x.shape[2]
    if x.shape[3] != 32 or not conv_s12_supported(freq, cout):
        raise CtcAsrError('conv_s12_fwd16: unsupported layer shape {} -> {} channels'.format(
            tuple(x.shape), cout))
    shape = (frames, batch, freq // 2, cout) if time_major else (batch, frames, freq // 2, cout)
    out = torch.empty(shape, dtype=torch.float32, device=x.device) if out is None else out
    with _Timed('conv_s12_fwd'):
        _check(load().ctcasr_conv_s12_fwd16(_dev(x, name='x'), float(x_scale),
                                            _dev(packed16, torch.uint8, 'packed16'),
                                            _dev(bias, name='bias'), _dev(out, name='y'), batch,
                                            frames, freq, cout, float(relu_cutoff),
                                            1 if time_major else 0, _stream()), 'conv_s12_fwd16')
    return out


@_on_tensor_device
def conv_s12_bwd_data(dz, packed, out=None, time_major=False, act=None, relu_cutoff=0.0):
    """dz f32[B,T,F/2,cout] (NHWC; ``time_major``: [T,B,F/2,cout]) -> dx f32[B,T,F,32].
    ``act`` (the layer's stored output, same layout): dz is the gradient w.r.t. that output and
    the mask of min(max(., 0), relu_cutoff) is applied while dz is staged."""
    if time_major:
        frames, batch, freq_out, cout = dz.shape
    else:
        batch, frames, freq_out, cout = dz.shape
    if not conv_s12_supported(2 * freq_out, cout):
        raise CtcAsrError('conv_s12_bwd_data: unsupported layer shape {}'.format(tuple(dz.shape)))
    out = torch.empty((batch, frames, 2 * freq_out, 32), dtype=torch.float32, device=dz.device) \
        if out is None else out
    with _Timed('conv_s12_bwd_data'):
        _check(load().ctcasr_conv_s12_bwd_data(_dev(dz, name='dz'), _dev(packed, name='packed'),
                                               _dev(out, name='dx'), batch, frames,
                                               2 * freq_out, cout, 1 if time_major else 0,
                                               _dev(act, name='act'), float(relu_cutoff),
                                               _stream()), 'conv_s12_bwd_data')
    return out


@_on_tensor_device
def conv_s12_bwd_data16(dz, packed16, out=None, time_major=False, act=None, relu_cutoff=0.0):
    """`conv_s12_bwd_data` with its products on the fp16 matrix pipe: every dz cell (frame,
    position) is scaled by its own power of two while it is staged - no bound on dz is assumed.
    ``packed16`` from `conv_s12_pack_weights16`."""
    if time_major:
        frames, batch, freq_out, cout = dz.shape
    else:
        batch, frames, freq_out, cout = dz.shape
    if not conv_s12_supported(2 * freq_out, cout):
        raise CtcAsrError('conv_s12_bwd_data16: unsupported layer shape {}'.format(tuple(dz.shape)))
    out = torch.empty((batch, frames, 2 * freq_out, 32), dtype=torch.float32, device=dz.device) \
        if out is None else out
    with _Timed('conv_s12_bwd_data'):
        _check(load().ctcasr_conv_s12_bwd_data16(_dev(dz, name='dz'),
                                                 _dev(packed16, torch.uint8, 'packed16'),
                                                 _dev(out, name='dx'), batch, frames,
                                                 2 * freq_out, cout, 1 if time_major else 0,
                                                 _dev(act, name='act'), float(relu_cutoff),
                                                 _stream()), 'conv_s12_bwd_data16')
    return out


@_on_tensor_device
def conv_s12_wrw(dz, x, out=None, time_major=False, act=None, relu_cutoff=0.0, dbias=None):
    """Kernel gradient of the same layer: dz f32[B,T,F/2,cout] (``time_major``: [T,B,F/2,cout]),
    x f32[B,T,F,32] (NHWC) -> dw f32[cout,32,11,21].  ``act`` / ``relu_cutoff``: see
    `conv_s12_bwd_data`; ``dbias`` f32[cout] (zeroed by the caller) then receives the bias
    gradient, the column sums of the masked dz."""
    if time_major:
        frames, batch, freq_out, cout = dz.shape
    else:
        batch, frames, freq_out, cout = dz.shape
    if tuple(x.shape) != (batch, frames, 2 * freq_out, 32) or \
            not conv_s12_supported(2 * freq_out, cout):
        raise CtcAsrError('conv_s12_wrw: unsupported layer shape {} / {}'.format(
            tuple(dz.shape), tuple(x.shape)))
    out = torch.empty((cout, 32, 11, 21), dtype=torch.float32, device=x.device) if out is None \
        else out
    workspace = _workspace(load().ctcasr_conv_s12_wrw_workspace_bytes(batch, frames,
                                                                      2 * freq_out, cout),
                           x.device)
    with _Timed('conv_s12_wrw'):
        _check(load().ctcasr_conv_s12_wrw(_dev(dz, name='dz'), _dev(x, name='x'),
                                          _dev(out, name='dw'), batch, frames, 2 * freq_out, cout,
                                          1 if time_major else 0, _dev(act, name='act'),
                                          float(relu_cutoff), _dev(dbias, name='dbias'),
                                          _dev(workspace, torch.uint8, 'workspace'),
                                          workspace.numel(), _stream()), 'conv_s12_wrw')
    return out


@_on_tensor_device
def conv_s12_wrw16(dz, x, x_scale, out=None, time_major=False, act=None, relu_cutoff=0.0,
                   dbias=None):
    """`conv_s12_wrw` with its products on the fp16 matrix pipe: for x with a known bound,
    bound * x_scale < 65504 (``x_scale`` a power of two); dz is scaled per output channel on the
    device."""
    if time_major:
        frames, batch, freq_out, cout = dz.shape
    else:
        batch, frames, freq_out, cout = dz.shape
    if tuple(x.shape) != (batch, frames, 2 * freq_out, 32) or \
            not conv_s12_supported(2 * freq_out, cout):
        raise CtcAsrError('conv_s12_wrw16: unsupported layer shape {} / {}'.format(
            tuple(dz.shape), tuple(x.shape)))
    out = torch.empty((cout, 32, 11, 21), dtype=torch.float32, device=x.device) if out is None \
        else out
    workspace = _workspace(load().ctcasr_conv_s12_wrw16_workspace_bytes(batch, frames,
                                                                        2 * freq_out, cout),
                           x.device)
    with _Timed('conv_s12_wrw'):
        _check(load().ctcasr_conv_s12_wrw16(_dev(dz, name='dz'), _dev(x, name='x'),
                                            float(x_scale), _dev(out, name='dw'), batch, frames,
                                            2 * freq_out, cout, 1 if time_major else 0,
                                            _dev(act, name='act'), float(relu_cutoff),
                                            _dev(dbias, name='dbias'),
                                            _dev(workspace, torch.uint8, 'workspace'),
                                            workspace.numel(), _stream()), 'conv_s12_wrw16')
    return out


@_on_tensor_device
def conv0_fwd(x, weight, bias=None, out=None, relu_cutoff=0.0):
    """First DS2 convolution: x f32[B,T,80] -> f32[B,ceil(T/2),40,32] (NHWC); weight
    f32[32,1,11,41], stride (2,2), TensorFlow SAME padding; ``relu_cutoff`` > 0 fuses
    min(max(., 0), cutoff) into the epilogue."""
    batch, frames = x.shape[0], x.shape[1]
    if x.shape[2] != 80 or tuple(weight.shape) != (32, 1, 11, 41):
        raise CtcAsrError('conv0_fwd covers x [B,T,80] and w [32,1,11,41] only.')
    out = torch.empty((batch, (frames + 1) // 2, 40, 32), dtype=torch.float32,
                      device=x.device) if out is None else out
    with _Timed('conv0_fwd'):
        _check(load().ctcasr_conv0_fwd(_dev(x, name='x'), _dev(weight, name='weight'),
                                       _dev(bias, name='bias'), _dev(out, name='y'), batch,
                                       frames, float(relu_cutoff), _stream()), 'conv0_fwd')
    return out


@_on_tensor_device
def conv0_pack_weights16(weight, out=None):
    """weight f32[32, 1, 11, 41] -> uint8 buffer for `conv0_fwd16` (fp16 pieces in fragment order,
    scale found on the device); re-pack whenever the weights change."""
    if tuple(weight.shape) != (32, 1, 11, 41):
        raise CtcAsrError('conv0_pack_weights16 covers w [32,1,11,41] only.')
    nbytes = load().ctcasr_conv0_pack16_bytes()
    out = torch.empty(nbytes, dtype=torch.uint8, device=weight.device) if out is None else out
    _check(load().ctcasr_conv0_pack_weights16(_dev(weight, name='weight'),
                                              _dev(out, torch.uint8, 'packed16'), _stream()),
           'conv0_pack_weights16')
    return out


@_on_tensor_device
def conv0_fwd16(x, packed16, bias=None, out=None, relu_cutoff=0.0):
    """`conv0_fwd` with its products on the fp16 matrix pipe (no bound on x is assumed)."""
    batch, frames = x.shape[0], x.shape[1]
    if x.shape[2] != 80:
        raise CtcAsrError('conv0_fwd16 covers x [B,T,80] only.')
    out = torch.empty((batch, (frames + 1) // 2, 40, 32), dtype=torch.float32,
                      device=x.device) if out is None else out
    with _Timed('conv0_fwd'):
        _check(load().ctcasr_conv0_fwd16(_dev(x, name='x'), _dev(packed16, torch.uint8, 'packed16'),
                                         _dev(bias, name='bias'), _dev(out, name='y'), batch,
                                         frames, float(relu_cutoff), _stream()), 'conv0_fwd16')
    return out


@_on_tensor_device
def conv0_wrw16(dz, x, out=None, act=None, relu_cutoff=0.0, dbias=None):
    """`conv0_wrw` with its products on the fp16 matrix pipe."""
    batch, frames = x.shape[0], x.shape[1]
    if x.shape[2] != 80 or tuple(dz.shape) != (batch, (frames + 1) // 2, 40, 32):
        raise CtcAsrError('conv0_wrw16 covers x [B,T,80], dz [B,ceil(T/2),40,32] only.')
    out = torch.empty((32, 1, 11, 41), dtype=torch.float32, device=x.device) if out is None \
        else out
    workspace = _workspace(load().ctcasr_conv0_wrw16_workspace_bytes(batch, frames), x.device)
    with _Timed('conv0_wrw'):
        _check(load().ctcasr_conv0_wrw16(_dev(dz, name='dz'), _dev(x, name='x'),
                                         _dev(out, name='dw'), batch, frames,
                                         _dev(act, name='act'), float(relu_cutoff),
                                         _dev(dbias, name='dbias'),
                                         _dev(workspace, torch.uint8, 'workspace'),
                                         workspace.numel(), _stream()), 'conv0_wrw16')
    return out


@_on_tensor_device
def conv0_wrw(dz, x, out=None, act=None, relu_cutoff=0.0, dbias=None):
    """Kernel gradient of the first DS2 convolution: dz f32[B,ceil(T/2),40,32] (NHWC),
    x f32[B,T,80] -> dw f32[32,1,11,41]; ``act`` / ``relu_cutoff`` / ``dbias`` as in
    `conv_s12_wrw`."""
    batch, frames = x.shape[0], x.shape[1]
    if x.shape[2] != 80 or tuple(dz.shape) != (batch, (frames + 1) // 2, 40, 32):
        raise CtcAsrError('conv0_wrw covers x [B,T,80], dz [B,ceil(T/2),40,32] only.')
    out = torch.empty((32, 1, 11, 41), dtype=torch.float32, device=x.device) if out is None \
        else out
    workspace = _workspace(load().ctcasr_conv0_wrw_workspace_bytes(batch, frames), x.device)
    with _Timed('conv0_wrw'):
        _check(load().ctcasr_conv0_wrw(_dev(dz, name='dz'), _dev(x, name='x'),
                                       _dev(out, name='dw'), batch, frames, _dev(act, name='act'),
                                       float(relu_cutoff), _dev(dbias, name='dbias'),
                                       _dev(workspace, torch.uint8, 'workspace'),
                                       workspace.numel(), _stream()), 'conv0_wrw')
    return out


@_on_tensor_device
def transpose_batched(src, out=None):
    """src f32[N, R, C] -> out f32[N, C, R]."""
    batch, rows, cols = src.shape
    out = torch.empty((batch, cols, rows), dtype=torch.float32, device=src.device) \
        if out is None else out
    _check(load().ctcasr_transpose_batched(_dev(src, name='src'), _dev(out, name='out'), batch,
                                           rows, cols, _stream()), 'transpose_batched')
    return out


@_on_tensor_device
def adam_step(param, grad, m, v, step, lr=1e-5, beta1=0.9, beta2=0.999, epsilon=1e-8,
              grad_scale=1.0, skip=None):
    """TensorFlow-form Adam over the flat arenas.  ``skip`` (optional int32 device tensor): the
    update is dropped on the device when skip[0] != 0 (`step_guard`)."""
    _check(load().ctcasr_adam_step(_dev(param, name='param'), _dev(grad, name='grad'),
                                   _dev(m, name='m'), _dev(v, name='v'), param.numel(), float(lr),
                                   float(beta1), float(beta2), float(epsilon), int(step),
                                   float(grad_scale), _dev(skip, torch.int32, 'skip'), _stream()),
           'adam_step')


def rnn_timeout_words(cell, workspace, num_steps, batch, hidden):
    """Device addresses (ints; 0 = none) of the sticky time-out words of the persistent kernels'
    row blocks inside ``workspace`` - what `step_guard` reads."""
    out = []
    for block in range(2):
        off = load().ctcasr_rnn_timeout_word_offset(CELL_IDS[cell], int(num_steps), int(batch),
                                                    int(hidden), block)
        out.append(0 if off == ctypes.c_size_t(-1).value else workspace.data_ptr() + off)
    return out


@_on_tensor_device
def step_guard(status, per_utterance_loss, timeout_words=(0, 0), out=None, wgrad_word=True):
    """out int32[2]: [0] = 1 when this step's gradients must not be applied (a CTC status word
    != 0, a non-finite per-utterance loss, a persistent recurrence that timed out, a part of a
    `wgrad16_gemm` tile on this device that gave up waiting for its turn), [1] = the time-out words
    or-ed (bit 30: the weight-gradient word).  Everything stays on the device
    (`adam_step(skip=out)`)."""
    if out is None:
        out = torch.empty(2, dtype=torch.int32, device=status.device)
    sync = _WGRAD16_SYNC.get(status.device.index) if wgrad_word else None
    _check(load().ctcasr_step_guard(_dev(status, torch.int32, 'status'),
                                    _dev(per_utterance_loss, name='per_utterance_loss'),
                                    status.numel(), timeout_words[0] or None,
                                    timeout_words[1] or None,
                                    None if sync is None else sync.data_ptr(),
                                    _dev(out, torch.int32, 'out'), _stream()), 'step_guard')
    return out


def occupy_cus(workgroups, busy_us):
    """Diagnostic: `workgroups` workgroups hold their CUs for `busy_us` on the current stream (the
    CU footprint of a collective's ring kernels; `engine.GradientReducer(stand_in=...)`)."""
    _check(load().ctcasr_occupy_cus(int(workgroups), int(busy_us), _stream()), 'occupy_cus')


@_on_tensor_device
def collective_traffic(scratch_a, scratch_b, workgroups, busy_us, payload_bytes):
    """Diagnostic: `occupy_cus` with a ring all-reduce's memory traffic - 2 x payload_bytes of
    a[i] += b[i] streamed through the scratch pair, paced over busy_us (include/ctcasr.h)."""
    _check(load().ctcasr_collective_traffic(
        int(workgroups), int(busy_us), _dev(scratch_a, name='scratch_a'),
        _dev(scratch_b, name='scratch_b'), scratch_a.numel(), int(payload_bytes), _stream()),
        'collective_traffic')


@_on_tensor_device
def absmax(x, out):
    """out (one int32 word, zeroed by the caller) = max(out, bit pattern of max |x|)."""
    _check(load().ctcasr_absmax(_dev(x.reshape(-1), name='x'), x.numel(),
                                _dev(out, torch.int32, 'out'), _stream()), 'absmax')
    return out


_FEATURE_TABLES = {}


def features_num_frames(num_samples):
    return load().ctcasr_features_num_frames(int(num_samples))


@_on_tensor_device
def features(pcm, num_samples, feature_type='mel', normalization='local',
             drop_every_second_frame=False, sampling_rate=16000, out=None, out_len=None):
    """pcm int16[B, N] (device), num_samples int32[B] (device) -> (f32[B, T, 80], i32[B])."""
    if pcm.dtype != torch.int16:
        raise CtcAsrError('pcm must be int16 (raw WAV samples, not rescaled).')
    batch, max_samples = pcm.shape
    dev = pcm.device
    key = (dev.index, int(sampling_rate))
    if key not in _FEATURE_TABLES:
        tables = _workspace(load().ctcasr_features_tables_bytes(), dev)
        _check(load().ctcasr_features_init_tables(_dev(tables, torch.uint8, 'tables'),
                                                  int(sampling_rate), _stream()),
               'features_init_tables')
        _FEATURE_TABLES[key] = tables
    tables = _FEATURE_TABLES[key]
    frames = features_num_frames(max_samples)
    if drop_every_second_frame:
        frames = (frames + 1) // 2
    out = torch.empty((batch, frames, 80), dtype=torch.float32, device=dev) if out is None else out
    out_len = torch.empty(batch, dtype=torch.int32, device=dev) if out_len is None else out_len
    workspace = _workspace(load().ctcasr_features_workspace_bytes(batch, max_samples), dev)
    _check(load().ctcasr_features(
        _dev(pcm, torch.int16, 'pcm'), _dev(num_samples, torch.int32, 'num_samples'), batch,
        max_samples, {'mel': 0, 'mfcc': 1}[feature_type],
        {'none': 0, 'local': 1, 'local_scalar': 2}[normalization],
        1 if drop_every_second_frame else 0, _dev(tables, torch.uint8, 'tables'),
        _dev(out, name='out'), out.shape[1], _dev(out_len, torch.int32, 'out_len'),
        _dev(workspace, torch.uint8, 'workspace'), workspace.numel(), _stream()), 'features')
    return out, out_len
