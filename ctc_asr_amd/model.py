"""The CTC acoustic model: DS1 / DS2 stacks, CTC loss, decoding, scoring.

Drop-in counterpart of ``asr/model.py::CTCModel`` (``inference_fn`` :123-236, ``loss_fn``
:238-269, ``decode_fn`` :271-309, ``error_rates_fn`` :311-345) and of the layer helpers in
``asr/util/tf_contrib.py:34-194``.  The reference builds a TensorFlow graph and lets autodiff
and cuDNN do the rest; here the forward and backward passes are written out explicitly over
pre-allocated HBM buffers so that one training step is a fixed sequence of HIP launches
(capturable in a hipGraph), with the parameters, their gradients and the Adam moments living
in flat fp32 arenas (one fused optimiser launch, one contiguous all-reduce payload).

PyTorch supplies device memory, streams, the plain library GEMMs / convolutions
(hipBLASLt / MIOpen through ``torch.mm`` / ``aten.convolution``) and ``torch.distributed``;
the recurrence, CTC, softmax, activations epilogues, decoding and the optimiser are the HIP
kernels behind ``include/ctcasr.h``.
"""

import math
import os

import numpy as np
import torch

from ctc_asr_amd import hip, metrics, split_gemm
from ctc_asr_amd.labels import num_classes

CONV_KERNEL_SIZES = ((11, 41), (11, 21), (11, 21))   # (time, freq), asr/util/tf_contrib.py:66
CONV_STRIDES = ((2, 2), (1, 2), (1, 2))              # asr/util/tf_contrib.py:67
GATES = {'lstm': 4, 'gru': 3, 'rnn_relu': 1, 'rnn_tanh': 1}


class InfeasibleAlignmentError(ValueError):
    """Raised where ``tf.nn.ctc_loss`` raises InvalidArgumentError ("Not enough time for target
    transition sequence"), ``ignore_longer_outputs_than_inputs=False`` (``asr/model.py:259``)."""


class ModelConfig:
    """Network layout; field names are the reference's flag names (``asr/params.py``)."""

    def __init__(self, used_model='ds2', conv_filters=(32, 32, 96), num_units_dense=2048,
                 num_layers_rnn=4, num_units_rnn=2048, rnn_cell='rnn_relu', cudnn=True,
                 relu_cutoff=20.0, conv_dropout_rate=0.0, rnn_dropout_rate=0.0,
                 dense_dropout_rate=0.1, num_classes_=None, num_features=80, beam_width=1024):
        if used_model not in ('ds1', 'ds2'):
            raise ValueError('Unsupported model "{}" in flags.'.format(used_model))
        if rnn_cell not in GATES:
            raise ValueError('Unsupported rnn_cell "{}".'.format(rnn_cell))
        conv_filters = tuple(int(f) for f in conv_filters)
        if used_model == 'ds2' and not 1 <= len(conv_filters) <= len(CONV_STRIDES):
            raise ValueError('conv_layers(): Arguments filters, kernel_size, and strides must '
                             'contain the same number of elements.')
        self.used_model = used_model
        self.conv_filters = conv_filters
        self.num_units_dense = int(num_units_dense)
        self.num_layers_rnn = int(num_layers_rnn)
        self.num_units_rnn = int(num_units_rnn)
        self.rnn_cell = rnn_cell
        self.cudnn = bool(cudnn)
        self.relu_cutoff = float(relu_cutoff)
        self.conv_dropout_rate = float(conv_dropout_rate)
        self.rnn_dropout_rate = float(rnn_dropout_rate)
        self.dense_dropout_rate = float(dense_dropout_rate)
        self.num_classes = int(num_classes_ or num_classes())
        self.num_features = int(num_features)
        self.beam_width = int(beam_width)

    @classmethod
    def from_flags(cls, flags):
        return cls(used_model=flags.used_model, conv_filters=flags.conv_filters,
                   num_units_dense=flags.num_units_dense, num_layers_rnn=flags.num_layers_rnn,
                   num_units_rnn=flags.num_units_rnn, rnn_cell=flags.rnn_cell, cudnn=flags.cudnn,
                   relu_cutoff=flags.relu_cutoff, conv_dropout_rate=flags.conv_dropout_rate,
                   rnn_dropout_rate=flags.rnn_dropout_rate,
                   dense_dropout_rate=flags.dense_dropout_rate, num_classes_=flags.num_classes,
                   beam_width=flags.beam_width)

    @property
    def cell(self):
        """The cell that actually runs: ``cudnn=False`` ignores ``rnn_cell`` and uses the tanh
        BasicRNNCell (``asr/params.py:47-50`` TODO, ``asr/util/tf_contrib.py:185``)."""
        return self.rnn_cell if self.cudnn else 'rnn_tanh'

    def conv_output_freq(self):
        freq = self.num_features
        for _ in self.conv_filters:
            freq = -(-freq // 2)
        return freq

    def rnn_input_size(self):
        if self.used_model == 'ds2':
            return self.conv_output_freq() * self.conv_filters[-1]
        return self.num_units_dense

    def output_time(self, num_frames):
        return -(-num_frames // 2) if self.used_model == 'ds2' else num_frames


def same_padding(size, kernel, stride):
    """TensorFlow SAME padding: (out, pad_before, pad_after); the odd element goes to the end."""
    out = -(-size // stride)
    total = max((out - 1) * stride + kernel - size, 0)
    return out, total // 2, total - total // 2


def param_spec(cfg):
    """Ordered (name, shape) list in the *shared layout* (see ``oracle/nn.py``): conv kernels
    TensorFlow HWIO ``[kt, kf, Cin, Cout]``, dense kernels ``[in, out]``, RNN ``[2, G*H, I]``."""
    spec = []
    if cfg.used_model == 'ds2':
        c_in = 1
        for i, filt in enumerate(cfg.conv_filters):
            k_t, k_f = CONV_KERNEL_SIZES[i]
            spec.append(('conv{}/kernel'.format(i), (k_t, k_f, c_in, filt)))
            spec.append(('conv{}/bias'.format(i), (filt,)))
            c_in = filt
    else:
        d_in = cfg.num_features
        for i in range(3):
            spec.append(('dense{}/kernel'.format(i), (d_in, cfg.num_units_dense)))
            spec.append(('dense{}/bias'.format(i), (cfg.num_units_dense,)))
            d_in = cfg.num_units_dense
    gates, hidden = GATES[cfg.cell], cfg.num_units_rnn
    in_size = cfg.rnn_input_size()
    for i in range(cfg.num_layers_rnn):
        spec.append(('rnn{}/w_ih'.format(i), (2, gates * hidden, in_size)))
        spec.append(('rnn{}/w_hh'.format(i), (2, gates * hidden, hidden)))
        spec.append(('rnn{}/b_ih'.format(i), (2, gates * hidden)))
        spec.append(('rnn{}/b_hh'.format(i), (2, gates * hidden)))
        in_size = 2 * hidden
    spec.append(('dense4/kernel', (2 * hidden, cfg.num_units_dense)))
    spec.append(('dense4/bias', (cfg.num_units_dense,)))
    spec.append(('logits/kernel', (cfg.num_units_dense, cfg.num_classes)))
    spec.append(('logits/bias', (cfg.num_classes,)))
    return spec


def _truncated_normal(rng, shape, stddev):
    out = rng.normal(size=shape) * stddev
    bad = np.abs(out) > 2 * stddev
    while bad.any():
        out[bad] = rng.normal(size=int(bad.sum())) * stddev
        bad = np.abs(out) > 2 * stddev
    return out.astype(np.float32)


def init_params(cfg, seed=0):
    """Random initial parameters following the reference's initialisers: dense kernels
    truncated-normal(0.046875) (``asr/model.py:146``), conv kernels glorot-normal
    (``asr/util/tf_contrib.py:68``), cuDNN weights glorot-uniform per gate matrix and zero
    biases (``asr/model.py:208-211``).  (TensorFlow's RNG stream itself cannot be reproduced.)"""
    rng = np.random.default_rng(seed)
    params = {}
    gates, hidden = GATES[cfg.cell], cfg.num_units_rnn
    for name, shape in param_spec(cfg):
        if name.endswith('bias') or name.endswith('b_ih') or name.endswith('b_hh'):
            params[name] = np.zeros(shape, dtype=np.float32)
        elif name.startswith('conv'):
            k_t, k_f, c_in, c_out = shape
            std = math.sqrt(2.0 / (k_t * k_f * c_in + k_t * k_f * c_out))
            params[name] = _truncated_normal(rng, shape, std)
        elif name.startswith('rnn'):
            fan_in = shape[2]
            if cfg.cudnn:
                limit = math.sqrt(6.0 / (fan_in + hidden))
            else:   # one [I+H, H] BasicRNNCell kernel
                layer = int(name[3:name.index('/')])
                in_size = cfg.rnn_input_size() if layer == 0 else 2 * hidden
                limit = math.sqrt(6.0 / (in_size + hidden + hidden))
            params[name] = rng.uniform(-limit, limit, size=shape).astype(np.float32)
        else:
            params[name] = _truncated_normal(rng, shape, 0.046875)
    del gates
    return params


def to_oracle_layout(flat_params, cfg):
    """name->array dict  ->  the nested dict ``oracle/nn.py`` / ``oracle/torch_ref.py`` take."""
    out = {}
    if cfg.used_model == 'ds2':
        out['conv'] = [(flat_params['conv{}/kernel'.format(i)], flat_params['conv{}/bias'.format(i)])
                       for i in range(len(cfg.conv_filters))]
    else:
        out['dense'] = [(flat_params['dense{}/kernel'.format(i)],
                         flat_params['dense{}/bias'.format(i)]) for i in range(3)]
    out['rnn'] = [{k: flat_params['rnn{}/{}'.format(i, k)] for k in ('w_ih', 'w_hh', 'b_ih', 'b_hh')}
                  for i in range(cfg.num_layers_rnn)]
    out['dense4'] = (flat_params['dense4/kernel'], flat_params['dense4/bias'])
    out['logits'] = (flat_params['logits/kernel'], flat_params['logits/bias'])
    return out


def xw_pipeline_plan(t_out, chunks):
    """Schedule of the pipelined input projection (`CTCModel._rnn_fwd_pipelined`).

    The forward recurrence of a layer runs as ``chunks`` launches over the step ranges
    [bounds[c], bounds[c+1]); after launch c the forward direction's y is final for the TIMES
    [lo, hi) and the backward direction's for [T'-hi, T'-lo), and each contributes its half-K
    product to those rows of the next layer's xw.  Returns (bounds, plan): plan[c] is a list of
    (direction, t_begin, t_end, first) GEMMs, ``first`` = the rows are written
    (plain product, no read) instead of accumulated into.  Every time index is initialised exactly once and
    before anything is accumulated into it, for any T' and chunk count (the cuts are symmetric,
    so for even T' the two directions' ranges coincide and nothing is split)."""
    bounds = [t_out * c // chunks for c in range(chunks + 1)]
    for c in range(chunks + 1):
        if c > chunks - c:
            bounds[c] = t_out - bounds[chunks - c]
    started = np.zeros(t_out, dtype=bool)
    plan = []
    for c in range(chunks):
        lo, hi = bounds[c], bounds[c + 1]
        ops = []
        for d, (a, b) in enumerate(((lo, hi), (t_out - hi, t_out - lo))):
            run = a
            while run < b:
                end = run + 1
                while end < b and started[end] == started[run]:
                    end += 1
                ops.append((d, run, end, not started[run]))
                started[run:end] = True
                run = end
        plan.append(ops)
    return bounds, plan


class ParamArena:
    """Flat fp32 arenas in HBM for parameters, gradients and the two Adam moments.

    Compute layout differs from the shared layout only for conv kernels, which are stored
    ``[Cout, Cin, kt, kf]`` (what the convolution consumes); ``load`` / ``export`` convert.
    Every tensor starts on a 16-byte boundary.  Layers appear in forward order, so a layer's
    gradients are one contiguous slice (``layer_slices``) — the unit of the bucketed all-reduce.
    """

    def __init__(self, cfg, device):
        self.cfg, self.device = cfg, device
        self.offsets, self.shapes = {}, {}
        self.layer_slices = []          # [(layer_name, start, stop)] in forward order
        cursor, layer, layer_start = 0, None, 0
        for name, shape in param_spec(cfg):
            this_layer = name.split('/')[0]
            if this_layer != layer:
                if layer is not None:
                    self.layer_slices.append((layer, layer_start, cursor))
                layer, layer_start = this_layer, cursor
            if name.startswith('conv') and name.endswith('kernel'):
                shape = (shape[3], shape[2], shape[0], shape[1])
            self.offsets[name], self.shapes[name] = cursor, tuple(shape)
            cursor += (int(np.prod(shape)) + 3) // 4 * 4
        self.layer_slices.append((layer, layer_start, cursor))
        self.size = cursor
        self.param = torch.zeros(cursor, dtype=torch.float32, device=device)
        self.grad = torch.zeros(cursor, dtype=torch.float32, device=device)
        self.m = torch.zeros(cursor, dtype=torch.float32, device=device)
        self.v = torch.zeros(cursor, dtype=torch.float32, device=device)
        self.p = {n: self._view(self.param, n) for n in self.offsets}
        self.g = {n: self._view(self.grad, n) for n in self.offsets}
        self.version = 0                # bumped when the parameters are replaced wholesale

    def _view(self, arena, name):
        shape = self.shapes[name]
        start = self.offsets[name]
        return arena[start:start + int(np.prod(shape))].view(shape)

    def num_parameters(self):
        return sum(int(np.prod(s)) for s in self.shapes.values())

    def touch(self):
        """The parameters were replaced wholesale (a restore, a broadcast): whatever was derived
        from the previous ones - the lagged weight maxima of the fp16 range guard - is stale."""
        self.version += 1

    def load(self, flat_params):
        """Copy a name->array dict in the shared layout into the parameter arena."""
        for name in self.offsets:
            value = torch.as_tensor(np.asarray(flat_params[name]), dtype=torch.float32)
            if name.startswith('conv') and name.endswith('kernel'):
                value = value.permute(3, 2, 0, 1)
            self.p[name].copy_(value.contiguous().to(self.device))
        self.version += 1

    def export(self, which='param'):
        """name->numpy dict in the shared layout (``which``: 'param' or 'grad')."""
        views = self.p if which == 'param' else self.g
        out = {}
        for name, view in views.items():
            value = view.detach().cpu()
            if name.startswith('conv') and name.endswith('kernel'):
                value = value.permute(2, 3, 1, 0)
            out[name] = value.contiguous().numpy().copy()
        return out


def predicted_input_bounds(cfg, training):
    """Upper bounds of |input| of every recurrent layer and (last entry) of dense4 as
    `CTCModel.inference_fn` will find them (None: unbounded) - which decides, ahead of the forward
    pass, whether a layer's projections take the fp16 form (two pieces under a fixed scale) or the
    bf16 form: the front end's clipped ReLU (<= relu_cutoff, times 1 / keep_prob where a dropout
    follows), |h| <= 1 behind the gated / tanh cells, nothing behind the ReLU cell."""
    if cfg.used_model == 'ds2':
        bound = cfg.relu_cutoff / (1.0 - cfg.conv_dropout_rate)
    else:
        bound = cfg.relu_cutoff / (1.0 - (cfg.dense_dropout_rate if training else 0.0))
    rate = cfg.rnn_dropout_rate if training else 0.0
    bounds = []
    for i in range(cfg.num_layers_rnn):
        if rate > 0.0 and (i > 0 or not cfg.cudnn) and bound is not None:
            bound = bound / (1.0 - rate)
        bounds.append(bound)
        bound = None if cfg.cell == 'rnn_relu' else 1.0
        if rate > 0.0 and not cfg.cudnn and bound is not None:
            bound = bound / (1.0 - rate)
    return bounds + [bound]


class _WeightPieces:
    """Pieces of one weight matrix for the split GEMMs of a step.  ``fwd16`` / ``tr16``: fp16
    pieces (fixed scale `split_gemm.W_SCALE`) of the matrix / of its transpose, or None; the bf16
    pieces of either are built ahead only where the fp16 form does not apply and otherwise on
    first use (a layer whose input turns out unbounded, a backward pass after an evaluation-mode
    forward pass, dense4's gradients)."""

    def __init__(self, model, name, matrix, training):
        self.model, self.name, self.matrix, self.training = model, name, matrix, training
        self.fwd16 = self.tr16 = None
        self.dg16 = None        # packed for `hip.dgrad16_blockscaled` (instead of ``tr16``)
        self.rows_form = False  # fp16 pieces for an UNBOUNDED input (scales per row / column)
        self._fwd = self._tr = None

    def make_fwd(self, out=None):
        if self.name == 'dense4':       # K = the kernel's row axis: pieces stacked along the rows
            stacked = split_gemm.split_rows_stacked(self.matrix, split_gemm.B_ORDER, out=out)
            self._fwd = stacked.view(-1, self.matrix.shape[1])
        else:
            self._fwd = split_gemm.split(self.matrix, split_gemm.B_ORDER, out=out)
        return self._fwd

    def make_tr(self, out=None, scratch=None):
        if self.name == 'dense4':       # (its data gradient reads the kernel's own rows)
            self._tr = split_gemm.split(self.matrix, split_gemm.A_ORDER, out=out)
        else:
            rows, cols = self.matrix.shape
            if scratch is None:
                scratch = torch.empty((cols, rows), dtype=torch.float32,
                                      device=self.matrix.device)
            hip.transpose_batched(self.matrix.view(1, rows, cols), out=scratch.view(1, cols, rows))
            self._tr = split_gemm.split(scratch, split_gemm.A_ORDER, out=out)
        return self._tr

    def make_tr16(self, out=None, scratch=None):
        """fp16 pieces of the transpose (the library form of the data gradient)."""
        rows, cols = self.matrix.shape
        if scratch is None:
            scratch = torch.empty((cols, rows), dtype=torch.float32, device=self.matrix.device)
        hip.transpose_batched(self.matrix.view(1, rows, cols), out=scratch.view(1, cols, rows))
        self.tr16 = split_gemm.split16(scratch, split_gemm.W_SCALE, split_gemm.H_B, out=out)
        return self.tr16

    def __getitem__(self, index):
        if index == 0:
            return self._fwd if self._fwd is not None else self.make_fwd()
        if index == 1:
            return self._tr if self._tr is not None else self.make_tr()
        return (self.fwd16, self.tr16)[index - 2]


class CTCModel:
    """DS1 / DS2 acoustic model on one MI355X.  Mirrors the method surface of the reference's
    ``CTCModel`` with torch tensors in place of TensorFlow tensors; ``forward_backward`` /
    ``apply_gradients`` are the explicit counterparts of ``optimizer.minimize``."""

    def __init__(self, cfg, device='cuda', seed=0, params=None, conv_autotune=None):
        hip.load()
        # The reference's convolution stack (1 -> 32 -> 32 [-> 96] channels) runs entirely on this
        # package's own kernels - forward, data gradient and kernel gradient, any number of
        # frames, no padded copies - so variable-length batches cost nothing extra.  Other channel
        # counts (the small models of the tests) go through MIOpen on explicitly SAME-padded
        # inputs.  Its default "find" benchmarks every solver the first time a shape is seen (3-6 s
        # per new padded length), so the default there is the immediate (heuristic) mode;
        # `conv_autotune=True` / CTCASR_CONV_AUTOTUNE=1 opts fixed-shape runs into autotuning.
        if conv_autotune is None:
            conv_autotune = os.environ.get('CTCASR_CONV_AUTOTUNE', '0') == '1'
        if conv_autotune:
            torch.backends.cudnn.benchmark = True
        else:
            os.environ.setdefault('MIOPEN_FIND_MODE', 'FAST')
            torch.backends.cudnn.benchmark = False
        self.cfg = cfg
        self.device = torch.device(device)
        if self.device.type != 'cuda':
            raise hip.CtcAsrError('CTCModel runs on the MI355X only; there is no CPU path.')
        self.arena = ParamArena(cfg, self.device)
        self.arena.load(params if params is not None else init_params(cfg, seed))
        self.step_count = 0
        # weight-gradient GEMMs run on a low-priority side stream so that they fill the half of
        # the chip the latency-bound backward recurrence of the layer below leaves free
        self.overlap_wgrad = os.environ.get('CTCASR_OVERLAP_WGRAD', '1') == '1'
        # the reference's convolution stack runs on this package's own implicit-GEMM kernels
        # (forward, data gradient, kernel gradient; any T, no padded intermediates)
        self.own_conv = os.environ.get('CTCASR_OWN_CONV', '1') == '1'
        self._conv_packed = {}          # layer -> fragment-ordered weight copies
        self._conv_packed16 = {}        # ... and the fp16 pieces for the fp16-pipe forward kernel
        # forward pass of the 11 x 21 convolutions on the fp16 matrix pipe (bounded input)
        self.conv_f16 = os.environ.get('CTCASR_CONV_F16', '1') == '1'
        self.conv_wrw_f16 = os.environ.get('CTCASR_CONV_WRW_F16', '1') == '1'
        # backward of the conv epilogue (mask + bias gradient) inside the gradient kernels
        self.conv_fused_bwd = os.environ.get('CTCASR_CONV_FUSED_BWD', '1') == '1'
        # Side-stream work that should run BESIDE a half-chip persistent recurrence launch waits
        # behind a residency gate: the launch posts a ticket once all its workgroups hold their
        # CUs, a one-lane gate kernel on the side stream waits for it (at most this long; 0
        # disables the gates)
        self.side_gate_max_us = int(os.environ.get('CTCASR_SIDE_GATE_US', '300'))
        self._ticket = 0
        # launches per persistent backward recurrence: the weight-gradient GEMMs of the steps one
        # launch has finished run beside the next launch instead of queueing up behind the layer.
        # 0 = the default, 2 launches.  (Rounds 1 - 3 ran 3 above 16 rows: C3 97.7 ms per step
        # against 102.8 with 2 and 99.1 with 4 when the GEMMs were fp32.  With the fp16 GEMMs and
        # the fp16-pipe recurrence the side stream is as long as the recurrence itself and fewer,
        # bigger GEMMs win: C3 55.6 - 55.9 ms with 2 against 56.8 - 56.9 with 3, and 60 launches
        # fewer per step, profiles/r04_ab.md)
        self.bwd_chunks = max(0, int(os.environ.get('CTCASR_BWD_CHUNKS', '0')))
        # launches per persistent FORWARD recurrence when the next layer's input projection is
        # pipelined with it on the other half of the chip (1 = off: whole-chip single launch)
        self.fwd_chunks = max(1, int(os.environ.get('CTCASR_FWD_CHUNKS', '4')))
        # ... for batches up to this size: above 16 rows the forward kernel runs one group of
        # workgroups per 16-row tile and fills the chip whatever the flag says, so the projection
        # GEMMs would only compete with it (measured at B = 32: profiles/r03_gemm_...md)
        self.fwd_pipeline_max_batch = int(os.environ.get('CTCASR_FWD_PIPELINE_MAX_BATCH', '16'))
        # the big fp32 GEMMs (RNN input projections and their gradients) as three-piece bf16
        # splits on the bf16 matrix pipe: fp32-grade results (split_gemm.py), 1.2x the GEMM rate
        self.split_gemm = os.environ.get('CTCASR_SPLIT_GEMM', '1') == '1'
        self.split_dense4 = os.environ.get('CTCASR_SPLIT_DENSE4', '1') == '1'
        # forward projections of bounded layer inputs (behind a clipped ReLU, |h| <= 1) as TWO
        # fp16 pieces and three products: the fp32 GEMM's accuracy at half the bf16 form's cost
        self.fwd_f16 = os.environ.get('CTCASR_FWD_F16', '1') == '1'
        # ... and the gradient GEMMs of those layers in the same form, the unbounded operand (dxw)
        # scaled per column (weight gradients) / per row (data gradient) on the device
        self.bwd_f16 = os.environ.get('CTCASR_BWD_F16', '1') == '1'
        self.split_wgrad = os.environ.get('CTCASR_SPLIT_WGRAD', '1') == '1'
        # the data gradient dxw W_ih of an LSTM-1024 layer by the own block-scaled kernel that reads
        # the fp16 pieces the fp16-pipe backward recurrence published (csrc/dgrad16.hip): no row
        # split of dxw, no library GEMM on the main stream - and so no "one library GEMM at a
        # time" wait for the side stream in front of it
        self.own_dgrad = os.environ.get('CTCASR_OWN_DGRAD', '1') == '1'
        # the weight gradients of a recurrent layer (LSTM / plain cells, fixed-scale fp16 operands)
        # through the own kernel (csrc/wgrad16.hip) instead of the library's TN GEMMs
        self.own_wgrad = os.environ.get('CTCASR_OWN_WGRAD', '1') == '1'
        # workgroups a tile's row sum is cut into (192 tiles of a direction's W_ih + W_hh at
        # H = 1024 leave a quarter of the chip idle, or make 1.5 rounds on the half beside a
        # recurrence launch)
        # (2 since the staggered backward recurrence: the recurrence launches got shorter, the
        # weight gradients beside them are what the data-gradient kernel then shares the chip with -
        # C3 48.0 -> 47.3 ms per step on one box, 47.6 -> 47.0 on another, 47.6 -> 46.7 with the
        # layer input packed once for both directions; 3: the same; C2: no difference)
        self.own_wgrad_parts = int(os.environ.get('CTCASR_WGRAD_PARTS', '2'))
        # layers whose input has no bound (behind a ReLU-cell layer: the reference's default model)
        # in an fp16 form as well - the input split with a scale per ROW for the projection (as a
        # layer's dxw is for the data gradient), per COLUMN for the weight gradients
        # (`split_gemm.mm_rows16`, `ColScaled`) - instead of the bf16 form's six products
        self.unbounded_f16 = os.environ.get('CTCASR_UNBOUNDED_F16', '1') == '1'
        # ... and, being no library kernel, in halves: a finished launch of the backward recurrence
        # has published one direction's dgates for its range of time steps - that direction's share
        # of dx (half of the K axis) is multiplied beside the next launch ('side': on the side
        # stream ahead of the range's weight gradients, 'own': on a third stream), only the last
        # launch's share stays on the main stream ('0': the whole product behind the last launch)
        # (measured again with the staggered backward recurrence and 2-part weight-gradient
        # tiles: 'side' 46.58 / 46.79 ms against 46.56 / 46.78 for C3, 14.42 / 14.36 against
        # 14.23 / 14.19 for C2 - the plain form stays the default)
        self.dgrad_early = os.environ.get('CTCASR_DGRAD_EARLY', '0')
        self._dgrad_stream = None
        # the forward recurrence's own product h_(t-1) W_hh^T as two fp16 pieces per operand and
        # three products on the fp16 matrix pipe (LSTM-1024 persistent kernel; |h| <= 1, W_hh
        # scaled per workgroup inside the kernel) instead of fp32 MFMAs
        self.rnn_fwd_f16 = os.environ.get('CTCASR_RNN_FWD_F16', '1') == '1'
        # ... and the backward recurrence's dgates W_hh (LSTM-1024 persistent kernel): dgates scaled
        # per (producer workgroup, row) inside the kernel, which also hands the column maxima of
        # dxw to the fp16 weight-gradient GEMMs (no `colmax` pass over dxw)
        self.rnn_bwd_f16 = os.environ.get('CTCASR_RNN_BWD_F16', '1') == '1'
        # fp16-pipe kernels: direction 0 on XCDs 0 - 3, direction 1 on XCDs 4 - 7 - every exchange
        # block crosses the fabric into four L2s instead of eight (bit-identical results; backward
        # recurrence 8.56 -> 8.30 us per step alone, 9.9 -> 9.7 in the C3 step, the step 0.5 ms
        # shorter in two alternating A/Bs: profiles/r04_ab.md)
        self.rnn_xcd_flag = hip.RNN_XCD_SPLIT \
            if os.environ.get('CTCASR_RNN_XCD_SPLIT', '1') == '1' else 0
        # backward LSTM-1024 recurrence, 24 or 32 rows: the two 16-row tiles staggered by half a step
        # (prnn_bwd16s_kernel; bit-identical results) instead of both behind one barrier
        self.rnn_stagger_flag = hip.RNN_STAGGER \
            if os.environ.get('CTCASR_RNN_STAGGER', '1') == '1' else 0
        # ... with the K axis split over pairs of workgroups (prnn_bwd16k_kernel, round 6: built,
        # parity-green and 3.4 us per time step slower - profiles/r06_rnn_bwd_k_pair.md; off)
        if os.environ.get('CTCASR_RNN_KPAIR', '0') == '1':
            self.rnn_stagger_flag |= hip.RNN_KPAIR
        # LSTM-2048 backward recurrence (prnn_bwd16w_kernel): the K-pair form - there the loads are
        # bound by L2 throughput, not by a latency chain, and a pair fills a whole MFMA tile
        self.rnn_kpair_wide = os.environ.get('CTCASR_RNN_KPAIR_2048', '1') == '1'
        # the fp16-pipe forward kernel writes the fp16 pieces of its output itself (no split pass)
        self.rnn_fwd_pieces = os.environ.get('CTCASR_RNN_FWD_PIECES', '1') == '1'
        self._w_split, self._w_split_ready, self._w_split_bufs = {}, None, {}
        self._w_guard = None            # range check of the weights that go to fp16 (lagged)
        self._side_stream = None
        self.early_hooks = False        # see backward(); set by engine.Trainer
        # variant of the persistent backward recurrence (hip.RNN_*): default = 128 CUs, the
        # other half of the chip runs the weight-gradient GEMMs of the layer above
        self.rnn_bwd_flags = int(os.environ.get('CTCASR_RNN_BWD_FLAGS', str(hip.RNN_DEFAULT)))
        self._rnn_ws = {}               # (cell, B, H) -> (zero-initialised workspace, T')
        self.dropout_seed = int(seed) * 0x9E3779B1 + 1
        self._acts = None
        self._w_hh_t = [torch.empty((2, cfg.num_units_rnn, GATES[cfg.cell] * cfg.num_units_rnn),
                                    dtype=torch.float32, device=self.device)
                        for _ in range(cfg.num_layers_rnn)]

    # ------------------------------------------------------------------ residency tickets
    def _upcoming_ticket(self):
        """The ticket the NEXT persistent launch of this model will carry."""
        return self._ticket % 0xFFFFFF + 1

    def _take_ticket(self):
        self._ticket = self._upcoming_ticket()
        return self._ticket

    def _gate_side_stream(self, cell, workspace, t_out, batch, hidden):
        """With the side stream current: hold it until the next persistent launch (enqueued on
        the main stream right after this call) has all its workgroups running."""
        if self.side_gate_max_us > 0:
            hip.rnn_resident_gate(cell, workspace, t_out, batch, hidden, self._upcoming_ticket(),
                                  self.side_gate_max_us)

    # ------------------------------------------------------------------ pieces of the weights
    def _predicted_bounds(self, training):
        return predicted_input_bounds(self.cfg, training)

    def _weights_in_f16_range(self, names, views, side):
        """{name: bool}: may this weight matrix be scaled by the fixed `split_gemm.W_SCALE` into
        fp16 (its largest magnitude, times the scale, stays below half of fp16's range)?  The
        maxima are found on the device (`hip.absmax`, side stream) and travel to pinned host
        memory asynchronously: a step looks at the LATEST maxima that have arrived - weights move
        by O(learning rate) per step, the factor of two of head room covers the lag - and only the
        first step after the parameters were (re)loaded waits for its own.  An out-of-range
        matrix takes the bf16 form (fp32's exponent range); it never becomes inf."""
        fresh = self._w_guard is None or self._w_guard['version'] != self.arena.version or \
            self._w_guard['names'] != names
        if fresh:
            self._w_guard = {
                'version': self.arena.version, 'names': names, 'event': None,
                'dev': torch.zeros(len(names), dtype=torch.int32, device=self.device),
                'host': torch.zeros(len(names), dtype=torch.int32).pin_memory(),
                'max': None}
        guard = self._w_guard

        def measure():
            with torch.cuda.stream(side):
                guard['dev'].zero_()
                for k, view in enumerate(views):
                    hip.absmax(view, guard['dev'][k:k + 1])
                guard['host'].copy_(guard['dev'], non_blocking=True)
                guard['event'] = torch.cuda.Event()
                guard['event'].record(side)

        if fresh:
            measure()
            guard['event'].synchronize()
        if guard['event'] is not None and guard['event'].query():
            guard['max'] = guard['host'].view(torch.float32).clone().numpy()
            guard['event'] = None
        if guard['event'] is None:
            measure()                       # for a later step
        limit = 0.5 * split_gemm.F16_MAX / split_gemm.W_SCALE
        return {name: bool(guard['max'][k] <= limit) for k, name in enumerate(names)}

    def _prepare_weight_splits(self, rows, training, batch=None):
        """Pieces of the weights the split GEMMs of this step read (split_gemm.py).  Per recurrent
        layer whose input is bounded (every layer of the BASELINE configurations) and whose W_ih
        is in range: the fp16 pieces of W_ih [2GH, 3, in] for the forward projection and of its
        transpose for the data gradient; the bf16 pieces ([.., 6, ..]) only where the fp16 form
        does not apply - they are made on demand otherwise (`_WeightPieces`).  The dense4 kernel:
        stacked along its rows (forward), [in, 6, out] bf16 for its data gradient.  Built on the
        side stream while the front end runs (the weights are final since the last Adam step);
        `_weight_split` makes the main stream wait for them."""
        cfg, p = self.cfg, self.arena.p
        self._w_split, self._w_split_ready = {}, None
        if not self.split_gemm:
            return
        gh2 = 2 * GATES[cfg.cell] * cfg.num_units_rnn
        jobs = []
        for i in range(cfg.num_layers_rnn):
            w_ih = p['rnn{}/w_ih'.format(i)].view(gh2, -1)
            if split_gemm.worthwhile(rows, w_ih.shape[1], gh2):
                jobs.append((i, 'rnn{}'.format(i), w_ih))
        k4 = p['dense4/kernel']
        dense4 = self.split_dense4 and split_gemm.worthwhile(rows, k4.shape[0], k4.shape[1])
        if not jobs and not dense4:
            return
        main = torch.cuda.current_stream(self.device)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(self.device)
        side = self._side_stream
        bufs = self._w_split_bufs
        start = torch.cuda.Event()
        start.record(main)
        side.wait_event(start)              # the previous step is done with the buffers
        bounds = self._predicted_bounds(training)
        in_range = {}
        if self.fwd_f16:
            names = tuple(name for _, name, _ in jobs) + (('dense4',) if dense4 else ())
            views = [w for _, _, w in jobs] + ([k4] if dense4 else [])
            in_range = self._weights_in_f16_range(names, views, side)

        def rows_form(layer, name):
            # no bound on the input: fp16 pieces under scales found on the device (ReLU cells)
            return (self.fwd_f16 and self.unbounded_f16 and in_range.get(name, False) and
                    bounds[layer] is None)

        def f16_form(layer, name):
            return (self.fwd_f16 and in_range.get(name, False) and
                    split_gemm.f16_scale(bounds[layer]) is not None) or rows_form(layer, name)

        def buf(key, make):
            if key not in bufs:
                bufs[key] = make()
            return bufs[key]

        dense4_f16 = dense4 and f16_form(cfg.num_layers_rnn, 'dense4')
        # the layers whose data gradient the own block-scaled kernel will compute (it reads what
        # the fp16-pipe backward recurrence publishes): W_ih packed for it, no transposed pieces
        hidden = cfg.num_units_rnn
        own_dgrad = bool(
            training and batch and self.own_dgrad and self.bwd_f16 and self.rnn_bwd_f16 and
            cfg.cell == 'lstm' and cfg.cudnn and rows % batch == 0 and
            hip.dgrad16_supported(cfg.cell, rows // batch, batch, hidden) and
            hip.rnn_bwd_f16_supported(cfg.cell, rows // batch, batch, hidden,
                                      self.rnn_bwd_flags | hip.RNN_F16))
        with torch.cuda.stream(side):
            for i, name, w_ih in jobs:
                cols = w_ih.shape[1]
                pieces = _WeightPieces(self, name, w_ih, training)
                pieces.rows_form = rows_form(i, name)
                if f16_form(i, name):
                    pieces.fwd16 = split_gemm.split16(
                        w_ih, split_gemm.W_SCALE, split_gemm.H_B,
                        out=buf(name + '/f16', lambda: split_gemm.empty16(
                            gh2, cols, split_gemm.H_B, self.device)))
                    # the layer's gradient GEMMs take the fp16 form as well when its OUTPUT has
                    # fp16 pieces (the next layer's input / dense4's): else the bf16 pieces of
                    # the transpose are needed - made on demand
                    above_f16 = f16_form(i + 1, 'rnn{}'.format(i + 1)) \
                        if i + 1 < cfg.num_layers_rnn else dense4_f16
                    if training and self.bwd_f16 and above_f16 and own_dgrad:
                        pieces.dg16 = hip.dgrad16_pack_weights(
                            w_ih, hidden, split_gemm.W_SCALE,
                            out=buf(name + '/dg16', lambda: torch.empty(
                                hip.dgrad16_packed_bytes(cols), dtype=torch.uint8,
                                device=self.device)))
                    elif training and self.bwd_f16 and above_f16:
                        pieces.make_tr16(
                            buf(name + '/t16', lambda: split_gemm.empty16(
                                cols, gh2, split_gemm.H_B, self.device)),
                            buf(name + '/t', lambda: torch.empty(
                                (cols, gh2), dtype=torch.float32, device=self.device)))
                else:
                    pieces.make_fwd(buf(name + '/bf16', lambda: split_gemm.empty(
                        gh2, cols, split_gemm.B_ORDER, self.device)))
                    if training:
                        pieces.make_tr(
                            buf(name + '/tbf16', lambda: split_gemm.empty(
                                cols, gh2, split_gemm.A_ORDER, self.device)),
                            buf(name + '/t', lambda: torch.empty(
                                (cols, gh2), dtype=torch.float32, device=self.device)))
                self._w_split[name] = pieces
            if dense4:
                pieces = _WeightPieces(self, 'dense4', k4, training)
                pieces.rows_form = rows_form(cfg.num_layers_rnn, 'dense4')
                if dense4_f16:
                    stacked16 = buf('dense4/f16', lambda: torch.empty(
                        (3,) + tuple(k4.shape), dtype=torch.float16, device=self.device))
                    split_gemm.split16_rows_stacked(k4, split_gemm.W_SCALE, split_gemm.H_B,
                                                    out=stacked16)
                    pieces.fwd16 = stacked16.view(-1, k4.shape[1])
                else:
                    pieces.make_fwd(buf('dense4/bf16', lambda: torch.empty(
                        (6,) + tuple(k4.shape), dtype=torch.bfloat16, device=self.device)))
                if training and dense4_f16 and self.bwd_f16:
                    # data gradient dz K^T in the fp16 form (dz split with a scale per row on
                    # the device, like a recurrent layer's dxw): the kernel's own rows are the
                    # [N, 3, K] pieces it reads - no transpose
                    pieces.tr16 = split_gemm.split16(
                        k4, split_gemm.W_SCALE, split_gemm.H_B,
                        out=buf('dense4/t16', lambda: split_gemm.empty16(
                            k4.shape[0], k4.shape[1], split_gemm.H_B, self.device)))
                elif training:      # (bf16 form: six products)
                    pieces.make_tr(buf('dense4/tbf16', lambda: split_gemm.empty(
                        k4.shape[0], k4.shape[1], split_gemm.A_ORDER, self.device)), None)
                self._w_split['dense4'] = pieces
            self._w_split_ready = torch.cuda.Event()
            self._w_split_ready.record(side)

    def _weight_split(self, name, backward=False, acts=None):
        """The `_WeightPieces` of a weight, or None: fp32 GEMMs for this layer.  Indexable like the
        tuple it replaces: [0] bf16 pieces for the forward product, [1] for the data gradient,
        [2] / [3] the fp16 ones (None where the fp16 form does not apply); [0] and [1] are made
        on first use when they were not built ahead."""
        pieces, ready = (self._w_split, self._w_split_ready) if acts is None else \
            (acts['w_split'], acts['w_split_ready'])
        got = pieces.get(name)
        if got is None:
            return None
        if ready is not None:
            torch.cuda.current_stream(self.device).wait_event(ready)
        return got

    # ------------------------------------------------------------------ forward
    def _next_seed(self):
        self.dropout_seed = (self.dropout_seed * 6364136223846793005 + 1442695040888963407) \
            & 0xFFFFFFFFFFFFFFFF
        return self.dropout_seed

    def _dense_act(self, x2d, name, rate, training, xs=None):
        """x2d [rows, in] -> dropout(min(relu(x K + b), cutoff)); returns the activation.
        ``xs``: the bf16 pieces of x2d when the product runs as a split GEMM."""
        p = self.arena.p
        if xs is not None:
            out = split_gemm.mm_nn_stacked(xs, self._weight_split(name)[0])
        else:
            out = torch.mm(x2d, p[name + '/kernel'])
        hip.bias_act_fwd(out, p[name + '/bias'], self.cfg.relu_cutoff,
                         rate if training else 0.0, self._next_seed())
        return out

    def inference_fn(self, sequences, seq_length, training=True):
        """``sequences`` f32[B, T, 80], ``seq_length`` i32[B] -> (logits f32[T', B, C] time-major,
        seq_length i32[B]); keeps the activations for a following backward pass
        (``asr/model.py:123-236``)."""
        cfg, p = self.cfg, self.arena.p
        sequences = sequences.to(self.device, torch.float32).contiguous()
        batch, frames, _ = sequences.shape
        acts = {'training': training, 'batch': batch, 'conv_f16': {}}
        # A NaN / inf in an utterance's features makes that utterance's loss NaN in the reference
        # (TensorFlow's relu / minimum / matmul propagate it; NanTensorHook stops the run,
        # asr/model.py:368).  Here the clipped-ReLU epilogues and the saturating fp16 splits would
        # swallow it - finite loss, NaN weight gradients -, so the utterance is marked at the door
        # and `loss_fn` reports what TensorFlow would (three tiny kernels per step).
        acts['input_poison'] = torch.where(
            torch.isfinite(sequences).view(batch, -1).all(dim=1), 0.0, float('nan')).to(torch.float32)
        self._prepare_weight_splits(cfg.output_time(frames) * batch, training, batch)
        if cfg.used_model == 'ds2':
            # conv dropout: the reference never forwards `training` to conv_layers, so a
            # non-zero conv_dropout_rate fires in evaluation too (asr/model.py:161).
            # logical NCHW [B, 1, T, F], physically NHWC (channels_last) like the reference: the
            # implicit-GEMM convolution kernels are NHWC-native, no transposes around them
            x = sequences.unsqueeze(1).contiguous(memory_format=torch.channels_last)
            conv_in, conv_out, pads, own_kind = [], [], [], []
            layers = len(cfg.conv_filters)
            # The own kernels apply ReLU + min(., relu_cutoff) in their epilogue, and the LAST
            # layer then writes time-major [T', B, F', C] - what the recurrent stack reads -
            # instead of NHWC: no elementwise pass and no transpose between the stacks.  (With
            # conv dropout the epilogue kernel still runs: it owns the mask generator.)
            fused = cfg.conv_dropout_rate == 0.0
            last_time_major = False
            for i in range(layers):
                k_t, k_f = CONV_KERNEL_SIZES[i]
                s_t, s_f = CONV_STRIDES[i]
                _, pt0, pt1 = same_padding(x.shape[2], k_t, s_t)
                _, pf0, pf1 = same_padding(x.shape[3], k_f, s_f)
                own_kind.append('conv0' if self._own_conv0_layer(i, x.shape[3]) else
                                's12' if self._own_conv_layer(i, x.shape[3]) else None)
                cutoff = cfg.relu_cutoff if fused else 0.0
                if own_kind[i] == 'conv0':
                    # first layer: straight from the [B, T, 80] features, no padded copy
                    if self.conv_f16:
                        self._conv_packed16[0] = hip.conv0_pack_weights16(
                            p['conv0/kernel'], self._conv_packed16.get(0))
                        y = hip.conv0_fwd16(sequences, self._conv_packed16[0], p['conv0/bias'],
                                            relu_cutoff=cutoff).permute(0, 3, 1, 2)
                    else:
                        y = hip.conv0_fwd(sequences, p['conv0/kernel'], p['conv0/bias'],
                                          relu_cutoff=cutoff).permute(0, 3, 1, 2)
                    acts.setdefault('arithmetic_front', {})['conv0/forward'] = \
                        'fp16x3' if self.conv_f16 else 'fp32'
                    conv_in.append(None)
                    acts['features'] = sequences
                elif own_kind[i] == 's12':
                    # weights change every step: re-pack (2 x 946 KB), then one launch
                    kernel = p['conv{}/kernel'.format(i)]
                    last_time_major = fused and i == layers - 1
                    # the layer's input is the clipped ReLU of the layer before (fused epilogue
                    # or `bias_act_fwd`): bounded by relu_cutoff -> the forward product on the
                    # fp16 matrix pipe, two pieces per operand (csrc/conv16.hip)
                    x_scale = split_gemm.f16_scale(cfg.relu_cutoff) \
                        if (self.conv_f16 and fused and i > 0) else None
                    if x_scale is not None:
                        self._conv_packed16[i] = hip.conv_s12_pack_weights16(
                            kernel, self._conv_packed16.get(i))
                        y = hip.conv_s12_fwd16(x.permute(0, 2, 3, 1), x_scale,
                                               self._conv_packed16[i], kernel.shape[0],
                                               p['conv{}/bias'.format(i)], relu_cutoff=cutoff,
                                               time_major=last_time_major)
                    else:
                        self._conv_packed[i] = hip.conv_s12_pack_weights(
                            kernel, self._conv_packed.get(i))
                        y = hip.conv_s12_fwd(x.permute(0, 2, 3, 1), self._conv_packed[i],
                                             kernel.shape[0], p['conv{}/bias'.format(i)],
                                             relu_cutoff=cutoff, time_major=last_time_major)
                    acts.setdefault('arithmetic_front', {})['conv{}/forward'.format(i)] = \
                        'fp16x3' if x_scale is not None else 'fp32'
                    acts.setdefault('conv_f16', {})[i] = x_scale
                    if not last_time_major:
                        y = y.permute(0, 3, 1, 2)
                    conv_in.append(x)      # the own kernel gradient reads the plain input
                else:
                    xp = torch.nn.functional.pad(x, (pf0, pf1, pt0, pt1)) \
                        .contiguous(memory_format=torch.channels_last)
                    y = torch.ops.aten.convolution(xp, self._conv_kernel_cl(i),
                                                   p['conv{}/bias'.format(i)], [s_t, s_f],
                                                   [0, 0], [1, 1], False, [0, 0], 1)
                    y = y.contiguous(memory_format=torch.channels_last)
                    conv_in.append(xp)
                if own_kind[i] is None or not fused:
                    # elementwise epilogue on the NHWC storage ([B, T, F, C] view of the memory)
                    hip.bias_act_fwd(y.permute(0, 2, 3, 1), None, cfg.relu_cutoff,
                                     cfg.conv_dropout_rate, self._next_seed())
                conv_out.append(y)
                pads.append((pt0, pt1, pf0, pf1))
                x = y
            if last_time_major:            # y is [T', B, F', C] already
                t_out = x.shape[0]
                rnn_in = x.view(t_out, batch, -1)
            else:
                t_out = x.shape[2]
                # [B, C, T', F'] -> time-major [T', B, F'*C] (freq-major, channel-minor like NHWC)
                rnn_in = x.permute(2, 0, 3, 1).reshape(t_out, batch, -1)
            # (which layers ran on the own kernels is decided HERE, once: backward reads it back)
            acts.update(conv_in=conv_in, conv_out=conv_out, pads=pads, conv_own=own_kind,
                        last_time_major=last_time_major)
            seq_length = torch.full((batch,), t_out, dtype=torch.int32, device=self.device)
            # every element of the front end's output lies in [0, bound] (clip, then the dropout
            # scale): what the fp16 split of the first projection's input relies on
            in_bound = cfg.relu_cutoff / (1.0 - cfg.conv_dropout_rate)
        else:
            t_out = frames
            x2d = sequences.transpose(0, 1).reshape(t_out * batch, -1)
            dense_in, dense_out = [], []
            for i in range(3):
                dense_in.append(x2d)
                x2d = self._dense_act(x2d, 'dense{}'.format(i), cfg.dense_dropout_rate, training)
                dense_out.append(x2d)
            rnn_in = x2d.view(t_out, batch, -1)
            in_bound = cfg.relu_cutoff / (1.0 - (cfg.dense_dropout_rate if training else 0.0))
            acts.update(dense_in=dense_in, dense_out=dense_out)
            seq_length = torch.as_tensor(seq_length).to(self.device, torch.int32).contiguous()

        cell, hidden, gates = cfg.cell, cfg.num_units_rnn, GATES[cfg.cell]
        rnn_len = None if cfg.cudnn else seq_length
        layer_in, layer_out, reserves, drop_seeds = [], [], [], []
        arithmetic = acts['arithmetic'] = dict(acts.get('arithmetic_front', {}))
        # h W_hh^T of the forward recurrence on the fp16 matrix pipe (persistent LSTM / GRU kernels)
        f16_rec = (self.rnn_fwd_f16 and cell in ('lstm', 'gru') and
                   hip.rnn_persistent_supported(cell, t_out, batch, hidden))
        rnn_flags = (hip.RNN_F16 | self.rnn_xcd_flag) if f16_rec else hip.RNN_DEFAULT
        rnn_form = 'fp16x3' if f16_rec else 'fp32'
        in_split = []                   # bf16 pieces of the layer inputs (None: fp32 GEMM)
        in_split16 = []                 # (fp16 pieces, scale) where the forward used them
        pipelined_xw = None
        y16_of = None               # (tensor, its fp16 pieces, their scale) from a forward kernel
        x = rnn_in.contiguous()
        workspace = self._rnn_workspace(cell, t_out, batch, hidden)
        rnn_rate = cfg.rnn_dropout_rate if training else 0.0
        for i in range(cfg.num_layers_rnn):
            # dropout placement: cuDNN drops the input of layers 2..L (asr/model.py:203); the
            # DropoutWrapper of the BasicRNNCell path drops every cell's input AND output
            # (asr/util/tf_contrib.py:190-194)
            seeds = [None, None]
            if rnn_rate > 0.0 and (i > 0 or not cfg.cudnn):
                seeds[0] = self._next_seed()
                x = hip.dropout(x, rnn_rate, seeds[0])
                in_bound = in_bound / (1.0 - rnn_rate) if in_bound is not None else None
            w_ih = p['rnn{}/w_ih'.format(i)].view(2 * gates * hidden, -1)
            # biases that are plain additive terms (LSTM / RNN: both vectors; GRU: everything but
            # the recurrent bias of the candidate gate) are added to xw INSIDE the recurrence
            # kernel (`xw_bias`): the GEMM then is a plain product without a bias epilogue
            # (3.93 instead of 4.16 ms per layer at C3) and no pass over xw is spent on it
            xs = ws = pieces16 = None
            form = 'fp32'
            if pipelined_xw is not None:       # built beside the previous layer's recurrence
                xw, pipelined_xw = pipelined_xw, None
            elif self._weight_split('rnn{}'.format(i)) is not None:
                w_pieces = self._weight_split('rnn{}'.format(i))
                scale = split_gemm.f16_scale(in_bound) if w_pieces[2] is not None else None
                if scale is None and w_pieces[2] is not None and w_pieces.rows_form:
                    # no bound (behind a ReLU cell): a scale per row for this product, per column
                    # (made in the backward pass, on the side stream) for the weight gradients
                    x2d = x.view(t_out * batch, -1)
                    xw = split_gemm.mm_rows16(x2d, w_pieces[2], split_gemm.W_SCALE)
                    pieces16 = split_gemm.ColScaled(x2d)
                    form = 'fp16x3 (row / column scales)'
                elif scale is not None:
                    # bounded input: two fp16 pieces, three products (the bf16 pieces the weight
                    # gradients want are made in the backward pass, on the side stream); the
                    # pieces of a recurrent layer's output come out of its fp16-pipe forward
                    # kernel (`y16_of`), everything else is split here
                    if y16_of is not None and y16_of[0] is x and y16_of[2] == scale:
                        x16 = y16_of[1]
                    else:
                        x16 = split_gemm.split16(x.view(t_out * batch, -1), scale, split_gemm.H_A)
                    xw = split_gemm.mm_nt16(x16, w_pieces[2], scale * split_gemm.W_SCALE)
                    pieces16 = (x16, scale)
                    form = 'fp16x3'
                else:
                    xs = split_gemm.split(x.view(t_out * batch, -1), split_gemm.A_ORDER)
                    xw = split_gemm.mm_nt(xs, w_pieces[0])
                    form = 'bf16x6'
            else:
                xw = torch.mm(x.view(t_out * batch, -1), w_ih.t())
            in_split.append(xs)
            in_split16.append(pieces16)
            arithmetic['rnn{}/input_projection'.format(i)] = form
            if self._pipeline_forward(i, cell, t_out, batch, hidden, rnn_len, rnn_rate):
                y16_of = None
                y, reserve, workspace, pipelined_xw = self._rnn_fwd_pipelined(
                    i, xw, t_out, batch, hidden, gates, workspace)
            else:
                # the fp16-pipe kernel can hand the fp16 pieces of its output to whoever multiplies
                # them next (the next layer's projection / dense4; this layer's dW_hh): worth it
                # when that consumer takes the fp16 form at the kernel's own scale (|h| <= 1, no
                # dropout in between) and every row runs all steps
                y16 = None
                consumer = 'rnn{}'.format(i + 1) if i + 1 < cfg.num_layers_rnn else 'dense4'
                if (f16_rec and self.rnn_fwd_pieces and rnn_len is None and
                        not (rnn_rate > 0.0 and (not cfg.cudnn or i + 1 < cfg.num_layers_rnn)) and
                        hip.rnn_fwd_f16_supported(cell, t_out, batch, hidden, rnn_flags)):
                    pieces = self._weight_split(consumer)
                    if pieces is not None and pieces[2] is not None:
                        y16 = split_gemm.empty16(t_out * batch, 2 * hidden, split_gemm.H_A,
                                                 self.device)
                y, reserve, workspace = hip.rnn_fwd(
                    cell, xw.view(t_out, batch, 2, gates * hidden), p['rnn{}/w_hh'.format(i)],
                    rnn_len, b_hh_n=p['rnn{}/b_hh'.format(i)] if cell == 'gru' else None,
                    workspace=workspace, xw_bias=self._rnn_bias(i), flags=rnn_flags,
                    y16=None if y16 is None else y16.buf)
                y16_of = None if y16 is None else (y, y16, split_gemm.RNN_F16_H_SCALE)
            arithmetic['rnn{}/recurrence_fwd'.format(i)] = rnn_form
            layer_in.append(x)
            layer_out.append(y)
            reserves.append(reserve)
            x = y
            # |h| <= 1 for the gated cells and tanh; the ReLU cell's output has no bound
            in_bound = None if cell == 'rnn_relu' else 1.0
            if rnn_rate > 0.0 and not cfg.cudnn:
                seeds[1] = self._next_seed()
                x = hip.dropout(x, rnn_rate, seeds[1])
                in_bound = in_bound / (1.0 - rnn_rate) if in_bound is not None else None
            drop_seeds.append(seeds)
        acts.update(layer_in=layer_in, layer_out=layer_out, reserves=reserves, rnn_ws=workspace,
                    rnn_len=rnn_len, t_out=t_out, drop_seeds=drop_seeds, rnn_rate=rnn_rate,
                    in_split=in_split, out_split=[None] * cfg.num_layers_rnn,
                    in_split16=in_split16)

        rnn_flat = x.view(t_out * batch, 2 * hidden)
        flat_split, k4_pieces = None, self._weight_split('dense4')
        acts['flat16'] = None
        flat_scale = split_gemm.f16_scale(in_bound) \
            if k4_pieces is not None and k4_pieces[2] is not None else None
        if flat_scale is None and k4_pieces is not None and k4_pieces[2] is not None and \
                k4_pieces.rows_form:
            acts['flat16'] = split_gemm.ColScaled(rnn_flat)
            dense4 = split_gemm.mm_rows16(rnn_flat, k4_pieces[2], split_gemm.W_SCALE,
                                          stacked=True)
            arithmetic['dense4'] = 'fp16x3 (row / column scales)'
            hip.bias_act_fwd(dense4, p['dense4/bias'], cfg.relu_cutoff,
                             cfg.dense_dropout_rate if training else 0.0, self._next_seed())
        elif flat_scale is not None:
            # bounded input (the recurrent stack's output): fp16 pieces, three products
            if y16_of is not None and y16_of[0] is x and y16_of[2] == flat_scale:
                flat16 = y16_of[1]
            else:
                flat16 = split_gemm.split16(rnn_flat, flat_scale, split_gemm.H_A)
            acts['flat16'] = (flat16, flat_scale)
            dense4 = split_gemm.mm_nn16_stacked(flat16, k4_pieces[2],
                                                flat_scale * split_gemm.W_SCALE)
            arithmetic['dense4'] = 'fp16x3'
            hip.bias_act_fwd(dense4, p['dense4/bias'], cfg.relu_cutoff,
                             cfg.dense_dropout_rate if training else 0.0, self._next_seed())
        else:
            if k4_pieces is not None:
                # (the top layer's output pieces also serve its recurrent weight gradient)
                flat_split = split_gemm.split(rnn_flat, split_gemm.A_ORDER)
            dense4 = self._dense_act(rnn_flat, 'dense4', cfg.dense_dropout_rate, training,
                                     xs=flat_split)
            arithmetic['dense4'] = 'bf16x6' if flat_split is not None else 'fp32'
        acts.update(flat_split=flat_split, flat_of=x, flat_uses_split=k4_pieces is not None)
        logits = torch.mm(dense4, p['logits/kernel'])
        hip.bias_act_fwd(logits, p['logits/bias'], 0.0)
        # (backward reads the pieces and the per-layer GEMM forms of THIS forward pass back from
        # the activations, not from whatever a later forward pass of another shape / mode left)
        acts.update(rnn_flat=rnn_flat, dense4=dense4, w_split=self._w_split,
                    w_split_ready=self._w_split_ready)
        self._acts = acts
        logits = logits.view(t_out, batch, cfg.num_classes)
        self.last_logits, self.last_seq_length = logits, seq_length     # for logging / summaries
        return logits, seq_length

    def arithmetic(self):
        """What the GEMM-shaped work of the LAST forward pass multiplied in (tensors and
        accumulation are fp32 everywhere): per layer 'fp16x3' / 'bf16x6' (pieces of the fp32
        operands on the 16-bit matrix pipe, split_gemm.py) or 'fp32'."""
        acts = self._acts or {}
        return dict(acts.get('arithmetic', {}))

    def _rnn_workspace(self, cell, t_out, batch, hidden):
        """The recurrence workspace shared by every layer and pass of this model: zero-filled
        when created (the persistent kernels' time-out word is sticky - a launch never clears
        it, `check_rnn_error` reads and clears it), kept across steps and re-created only when
        the batch changes or a longer sequence needs a bigger exchange buffer."""
        need = hip.rnn_workspace_bytes(cell, t_out, batch, hidden)
        key = (cell, batch, hidden)
        have = self._rnn_ws.get(key)
        if have is None or have[0].numel() < need:
            if have is not None:        # do not lose a time-out recorded in the old buffer
                hip.rnn_poll_error(cell, have[0], have[1], batch, hidden)
            have = (hip.rnn_workspace(cell, t_out, batch, hidden, self.device), t_out)
            self._rnn_ws[key] = have
        return have[0]

    def check_rnn_error(self):
        """Raise `hip.CtcAsrError` (CTCASR_ERR_TIMEOUT) if a persistent recurrence kernel of
        any layer, pass or step since the last check gave up at a grid barrier: its output -
        and every gradient computed from it - is invalid.  Synchronises the stream."""
        for (cell, batch, hidden), (workspace, t_out) in self._rnn_ws.items():
            hip.rnn_poll_error(cell, workspace, t_out, batch, hidden)
        # ... or a part of a weight-gradient tile stopped waiting for its turn (its sum was not
        # added; the guard word has dropped the update)
        hip.wgrad16_check(self.device)

    def _pipeline_forward(self, layer, cell, t_out, batch, hidden, rnn_len, rnn_rate):
        """Whether layer ``layer``'s forward recurrence runs on half of the chip in step ranges
        with the NEXT layer's input projection on the other half: pays when that projection is a
        big GEMM (another LSTM layer follows), the steps map to the same time index for every row
        and nothing (dropout) sits between the layers."""
        if self._weight_split('rnn{}'.format(layer + 1)) is not None:
            # the split projection of the next layer is a 1.1 ms GEMM at B = 16: not worth slowing
            # this layer's recurrence to half of the chip for (C2: 21.8 against 22.3 ms per step)
            return False
        return (self.fwd_chunks > 1 and cell == 'lstm' and hidden == 1024 and
                batch <= self.fwd_pipeline_max_batch and
                rnn_len is None and
                rnn_rate == 0.0 and layer + 1 < self.cfg.num_layers_rnn and
                t_out >= 8 * self.fwd_chunks and
                hip.rnn_persistent_supported(cell, t_out, batch, hidden))

    def _rnn_fwd_pipelined(self, layer, xw, t_out, batch, hidden, gates, workspace):
        """Forward recurrence of ``layer`` in ``fwd_chunks`` launches on 128 CUs; after each launch
        the finished steps' share of the next layer's xw = y W_ih^T + b (split by direction:
        times [lo, hi) of the forward half of y, [T-hi, T-lo) of the backward half) is added on
        the side stream.  Returns (y, reserve, workspace, xw of the next layer)."""
        p, cell = self.arena.p, 'lstm'
        name, nxt = 'rnn{}'.format(layer), 'rnn{}'.format(layer + 1)
        main = torch.cuda.current_stream(self.device)
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(self.device)
        side = self._side_stream
        dev = xw.device
        gh = gates * hidden
        y = torch.empty((t_out, batch, 2 * hidden), dtype=torch.float32, device=dev)
        reserve = hip._workspace(hip.rnn_reserve_bytes(cell, t_out, batch, hidden), dev)
        xw_next = torch.empty((t_out * batch, 2 * gh), dtype=torch.float32, device=dev)
        w_next = p[nxt + '/w_ih'].view(2 * gh, 2 * hidden)
        bias_here = self._rnn_bias(layer)
        w_halves = [w_next[:, d * hidden:(d + 1) * hidden].t() for d in (0, 1)]
        bounds, plan = xw_pipeline_plan(t_out, self.fwd_chunks)

        def contribute(chunk):
            for d, run, end, first in plan[chunk]:
                rows = xw_next[run * batch:end * batch]
                src = y[run:end, :, d * hidden:(d + 1) * hidden] \
                    .reshape((end - run) * batch, hidden)
                if first:
                    torch.mm(src, w_halves[d], out=rows)
                else:
                    rows.addmm_(src, w_halves[d])

        chunks = self.fwd_chunks
        for c in range(chunks):
            lo, hi = bounds[c], bounds[c + 1]
            hip.rnn_fwd(cell, xw.view(t_out, batch, 2, gh), p[name + '/w_hh'], None, y=y,
                        reserve=reserve, workspace=workspace, steps=(lo, hi),
                        flags=hip.RNN_HALF_CHIP | ((hip.RNN_F16 | self.rnn_xcd_flag)
                                                   if self.rnn_fwd_f16 else 0),
                        xw_bias=bias_here, ticket=self._take_ticket())
            if c + 1 < chunks:
                ready = torch.cuda.Event()
                ready.record(main)
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    # these GEMMs run beside launch c + 1: let it take its 128 CUs first
                    self._gate_side_stream(cell, workspace, t_out, batch, hidden)
                    contribute(c)
            else:
                main.wait_stream(side)
                contribute(c)
        for tensor in (y, xw_next):
            tensor.record_stream(side)
        return y, reserve, workspace, xw_next

    def _own_conv0_layer(self, layer, freq_in):
        return (self.own_conv and layer == 0 and freq_in == 80 and
                tuple(self.arena.p['conv0/kernel'].shape) == (32, 1, 11, 41))

    def _own_conv_layer(self, layer, freq_in):
        kernel = self.arena.p['conv{}/kernel'.format(layer)]
        return (self.own_conv and tuple(kernel.shape[1:]) == (32, 11, 21) and
                CONV_STRIDES[layer] == (1, 2) and
                hip.conv_s12_supported(freq_in, kernel.shape[0]))

    def _conv_kernel_cl(self, layer):
        """Conv kernel [Cout, Cin, kt, kf] in channels_last memory (scratch copy per call).  The
        copy lives in a buffer with 1 KB of slack behind it: MIOpen's immediate-mode kernels for
        very small channel counts (the 4-channel test models) read a little past the end of the
        weight tensor, which is a GPU memory fault when that tensor happens to be the last one
        of an allocator segment."""
        weight = self.arena.p['conv{}/kernel'.format(layer)]
        c_out, c_in, k_t, k_f = weight.shape
        flat = torch.empty(weight.numel() + 256, dtype=torch.float32, device=weight.device)
        scratch = flat.as_strided((c_out, c_in, k_t, k_f),
                                  (k_t * k_f * c_in, 1, k_f * c_in, c_in))
        scratch.copy_(weight)
        return scratch

    def _rnn_bias(self, layer):
        """The bias folded into the input projection, one [2*G*H] vector (scratch): b_ih + b_hh,
        except for the GRU candidate gate whose recurrent bias sits inside r * (R_n h + b_Rn)."""
        p = self.arena.p
        b_ih, b_hh = p['rnn{}/b_ih'.format(layer)], p['rnn{}/b_hh'.format(layer)]
        if self.cfg.cell != 'gru':
            return (b_ih + b_hh).reshape(-1)
        hidden = self.cfg.num_units_rnn
        folded = b_ih + b_hh
        folded[:, 2 * hidden:] = b_ih[:, 2 * hidden:]
        return folded.reshape(-1)

    # ------------------------------------------------------------------ loss
    @staticmethod
    def pack_labels(labels, device):
        """Dense zero-padded ``[B, L]`` int labels (or a list of lists) -> concatenated ids,
        offsets, max length: the ``dense_to_sparse`` of ``asr/model.py:71`` (0 = pad/eos)."""
        if isinstance(labels, torch.Tensor):
            labels = labels.cpu().numpy()
        rows = [[int(v) for v in row if int(v) != 0] for row in labels]
        offsets = np.zeros(len(rows) + 1, dtype=np.int32)
        offsets[1:] = np.cumsum([len(r) for r in rows])
        flat = np.array([v for r in rows for v in r] or [0], dtype=np.int32)
        max_len = max([len(r) for r in rows] + [1])
        return (torch.as_tensor(flat).to(device), torch.as_tensor(offsets).to(device), max_len,
                rows)

    def loss_fn(self, logits, seq_length, labels, check=True):
        """Mean over the batch of the CTC loss (``asr/model.py:238-269``).  Also leaves
        d(mean loss)/d(logits) in ``self._acts['dlogits']`` for ``backward``.  Raises
        `InfeasibleAlignmentError` where TensorFlow raises (``check=False`` defers the
        device->host status read to the caller)."""
        if isinstance(labels, tuple) and len(labels) == 4 and torch.is_tensor(labels[0]):
            flat, offsets, max_len, _ = labels          # already packed by `pack_labels`
        else:
            flat, offsets, max_len, _ = self.pack_labels(labels, self.device)
        batch = logits.shape[1]
        per_utt, grad, status = hip.ctc_loss_fwd_bwd(logits, flat, offsets, seq_length, max_len,
                                                     grad_scale=1.0 / batch)
        if self._acts is not None:
            self._acts['dlogits'] = grad
            if self._acts.get('input_poison') is not None and \
                    self._acts['input_poison'].shape == per_utt.shape:
                per_utt = per_utt + self._acts['input_poison']
        self.last_status, self.last_per_utterance_loss = status, per_utt
        if check:
            self.check_status(status)
        return per_utt.mean()

    @staticmethod
    def check_status(status):
        bad = status.cpu().numpy()
        if (bad == 1).any():
            raise InfeasibleAlignmentError(
                'Not enough time for target transition sequence (batch rows {})'
                .format(np.nonzero(bad == 1)[0].tolist()))
        if (bad == 2).any():
            raise ValueError('CTC labels out of range or sequence_length > max_time (rows {})'
                             .format(np.nonzero(bad == 2)[0].tolist()))

    # ------------------------------------------------------------------ backward
    def backward(self, reduce_hook=None):
        """Back-propagate ``dlogits`` (set by `loss_fn`) through the stack; fills the gradient
        arena.  ``reduce_hook(layer_name, start, stop)`` is called as soon as a layer's slice of
        the arena is final (logits first, front-end last) — the bucketed all-reduce hook."""
        cfg, p, g, acts = self.cfg, self.arena.p, self.arena.g, self._acts
        if acts is None or 'dlogits' not in acts:
            raise RuntimeError('backward() needs inference_fn() and loss_fn() first.')
        slices = {name: (start, stop) for name, start, stop in self.arena.layer_slices}

        def done(layer):
            if reduce_hook is not None:
                reduce_hook(layer, *slices[layer])

        # (every weight gradient below ACCUMULATES into the arena - step ranges, directions and the
        # kernels' bias sums add their shares - so the arena starts from zero here)
        self.arena.grad.zero_()
        arith = acts.setdefault('arithmetic', {})
        training = acts['training']
        t_out, batch = acts['t_out'], acts['batch']
        hidden, gates, cell = cfg.num_units_rnn, GATES[cfg.cell], cfg.cell
        rows = t_out * batch
        dlogits = acts['dlogits'].view(rows, cfg.num_classes)
        main = torch.cuda.current_stream(self.device)
        side = main
        if self.overlap_wgrad:
            if self._side_stream is None:
                # (a CU-masked side stream was tried and measured slower: whatever it has left
                # when the recurrence ends keeps running on half of an otherwise idle chip)
                # HIP stream priorities (torch only offers default / high; a lowest-priority HIP
                # stream and a high-priority main stream were both tried) change nothing measurable
                self._side_stream = torch.cuda.Stream(self.device)
            side = self._side_stream
        deferred = []          # layer hooks that must wait for the side stream

        # The LSTM / GRU at H = 2048 run their backward recurrence on ALL 256 CUs (one direction
        # per launch): GEMMs on the side stream then only get in its way - every persistent
        # workgroup needs a CU of its own, the resident ones spin until the last GEMM workgroup
        # has drained (ref_best: 33.2 instead of 21.4 us per step, 187 instead of 177 ms per
        # training step).  Their weight gradients stay on the main stream; only the bottom
        # layer's overlap the front-end backward, where no persistent kernel runs.
        whole_chip_rnn = (cfg.cell in ('lstm', 'gru') and cfg.num_units_rnn == 2048 and
                          hip.rnn_persistent_supported(cfg.cell, acts['t_out'], acts['batch'],
                                                       cfg.num_units_rnn))

        persistent = hip.rnn_persistent_supported(cfg.cell, acts['t_out'], acts['batch'],
                                                  cfg.num_units_rnn)

        def on_side(tensors, fn, gate=False, beside_recurrence=True):
            """Run ``fn`` (weight-gradient work that nothing downstream in this backward pass
            reads) on the side stream once everything enqueued on the main stream so far is done.
            ``gate``: a persistent recurrence launch is enqueued next on the main stream; the
            side stream waits until its workgroups hold their 128 CUs before these GEMMs fill the
            chip (otherwise the recurrence starts only when the first GEMM has drained).
            ``beside_recurrence``: the work would run while a recurrence launch is on the main
            stream."""
            if side is main or (whole_chip_rnn and beside_recurrence):
                fn()
                return
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                if gate and persistent:
                    self._gate_side_stream(cfg.cell, acts['rnn_ws'], acts['t_out'], acts['batch'],
                                           cfg.num_units_rnn)
                fn()
            for tensor in tensors:       # blocks stay reserved until the side stream is done
                tensor.record_stream(side)

        # re-layouts of the recurrent weights for the backward kernels: off the critical path
        for i in range(cfg.num_layers_rnn):
            hip.transpose_batched(p['rnn{}/w_hh'.format(i)], out=self._w_hh_t[i])

        # logits layer
        torch.mm(acts['dense4'].t(), dlogits, out=g['logits/kernel'])
        hip.colsum_accumulate(dlogits, g['logits/bias'])
        d_dense4 = torch.mm(dlogits, p['logits/kernel'].t())
        done('logits')
        # dense4: dz and the data gradient are on the critical path, the kernel gradient is not
        dz = hip.bias_act_bwd(acts['dense4'], d_dense4, cfg.relu_cutoff,
                              cfg.dense_dropout_rate if training else 0.0, g['dense4/bias'])
        k4_split = self._weight_split('dense4', True, acts) if acts['flat_uses_split'] else None
        dz_split = None
        if k4_split is not None and k4_split[3] is not None:
            # (the kernel gradient's bf16 pieces of dz are made where it runs: the side stream)
            dy = split_gemm.dgrad16(dz, k4_split[3], split_gemm.W_SCALE) \
                .view(t_out, batch, 2 * hidden)
            acts['arithmetic']['dense4/data_gradient'] = 'fp16x3'
        elif k4_split is not None:
            dz_split = split_gemm.split(dz, split_gemm.B_ORDER)
            dy = split_gemm.mm_nt(dz_split, k4_split[1]).view(t_out, batch, 2 * hidden)
        else:
            dy = torch.mm(dz, p['dense4/kernel'].t()).view(t_out, batch, 2 * hidden)
        # `early_hooks` (opt-in, N > 1): a layer's hook - its bucket's all-reduce - fires on the
        # side stream right behind that layer's weight-gradient GEMMs instead of after the whole
        # backlog, so the collective hides behind the layers below (engine.Trainer)
        early = self.early_hooks

        # dense4's kernel gradient flat^T dz in the fp16 form (round 5; it was the last big bf16 x 6
        # product of the step, 1.46 ms on the side stream at C3): the forward pass's fp16 pieces of
        # the recurrent stack's output (fixed scale, or per-column scales behind a ReLU cell)
        # against dz scaled per column like a layer's dxw
        flat16 = acts.get('flat16')
        k4_f16 = (self.bwd_f16 and k4_split is not None and flat16 is not None and
                  self.split_wgrad and os.environ.get('CTCASR_DENSE4_WGRAD_F16', '1') == '1')
        acts['arithmetic']['dense4/kernel_gradient'] = \
            'fp16x3' if k4_f16 else ('bf16x6' if k4_split is not None else 'fp32')

        def dense4_weight_grad(dz_split=dz_split):
            if k4_f16:
                dz16, dz_inv = split_gemm.wgrad16_operand(dz)
                pieces, scale = flat16[0], flat16[1]
                inv = getattr(flat16, 'col_inv', None)
                if inv is None:
                    inv = torch.ones(pieces.cols, dtype=torch.float32, device=dz.device)
                g['dense4/kernel'].zero_()
                split_gemm.wgrad16(g['dense4/kernel'], pieces.buf, inv,
                                   split_gemm.Split(dz16, split_gemm.H_B), scale, 0,
                                   x_col_inv=dz_inv)
            elif k4_split is not None:
                if dz_split is None:
                    dz_split = split_gemm.split(dz, split_gemm.B_ORDER)
                if acts['flat_split'] is None:      # (the forward pass used fp16 pieces)
                    acts['flat_split'] = split_gemm.split(acts['rnn_flat'], split_gemm.A_ORDER)
                split_gemm.mm_tn_rows(g['dense4/kernel'], acts['flat_split'], dz_split, 0, rows,
                                      accumulate=False)
            else:
                torch.mm(acts['rnn_flat'].t(), dz, out=g['dense4/kernel'])
            if early:
                done('dense4')

        on_side([dz] + ([dz_split.buf] if dz_split is not None else []) +
                ([acts['flat_split'].buf] if acts['flat_split'] is not None else []),
                dense4_weight_grad, gate=True)
        if not early:
            deferred.append('dense4')

        # recurrent stack, top layer first
        need_dx_first = True
        rnn_rate = acts['rnn_rate']
        for i in range(cfg.num_layers_rnn - 1, -1, -1):
            name = 'rnn{}'.format(i)
            x, y = acts['layer_in'][i], acts['layer_out'][i]
            seeds = acts['drop_seeds'][i]
            if seeds[1] is not None:
                dy = hip.dropout(dy.contiguous(), rnn_rate, seeds[1])
            gh = gates * hidden
            # Cut the recurrence into launches when the steps map to the same time index for every
            # utterance (no per-row lengths: the cuDNN-semantics path) - dxw of a finished range
            # of steps is final, so its share of dW_ih / dW_hh starts on the side stream while the
            # next launch carries the recurrence on.
            # (Not for the LSTM / GRU at H = 2048: their backward kernels occupy the whole chip one
            # direction at a time, nothing could run beside them.)
            # GRU: dW_hh and db_hh are products / sums of drec (the candidate gate's gradient
            # scaled by r), which the backward kernel leaves in the reserve next to the gates.
            chunks = 1
            if (side is not main and acts['rnn_len'] is None and not whole_chip_rnn and
                    persistent):
                chunks = self.bwd_chunks or 2
                if t_out < 8 * chunks:
                    chunks = 1
            dy = dy.contiguous()
            dxw = torch.empty((t_out, batch, 2, gh), dtype=torch.float32, device=dy.device)

            drec = hip.rnn_gru_drec(acts['reserves'][i], t_out, batch, hidden) if cell == 'gru' \
                else dxw
            dxw2d = dxw.view(rows, 2 * gh)
            w_ih = p[name + '/w_ih'].view(2 * gh, -1)

            # bf16-split operands (split_gemm.py): the pieces of this layer's input and W_ih come
            # from the forward pass where it used them; y's are the next layer's input pieces
            w_pieces = self._weight_split(name, True, acts)
            use_split = w_pieces is not None
            # fp16 form of this layer's gradient GEMMs: when the forward pass left fp16 pieces of
            # the layer's input AND of its output (the next layer's input / dense4's), dxw is
            # split with per-column scales for the weight gradients (chunk by chunk) and with
            # per-row scales for the data gradient - no bf16 pieces of anything are needed
            x16 = acts['in_split16'][i]
            if i + 1 < cfg.num_layers_rnn:
                y16 = acts['in_split16'][i + 1] if acts['layer_in'][i + 1] is y else None
            else:
                y16 = acts['flat16'] if acts['flat_of'] is y else None
            g16 = (self.bwd_f16 and use_split and x16 is not None and y16 is not None and
                   (w_pieces[3] is not None or w_pieces.dg16 is not None) and
                   2 * gh <= 16384 and self.split_wgrad)
            ds = drs = None
            if use_split and not g16:
                ds = split_gemm.empty(rows, 2 * gh, split_gemm.B_ORDER, dy.device)
                drs = ds if cell != 'gru' else \
                    split_gemm.empty(rows, 2 * gh, split_gemm.B_ORDER, dy.device)

            def input_pieces(layer=i, x=x):
                # bf16 pieces of this layer's input: from the forward pass if it made them, else
                # made here - on the stream that runs the weight gradients - and kept: the layer
                # below reads its output's pieces (this very tensor's) from the same place
                if acts['in_split'][layer] is None:
                    acts['in_split'][layer] = split_gemm.split(x.view(rows, -1),
                                                               split_gemm.A_ORDER)
                return acts['in_split'][layer]

            def output_pieces(layer=i, y=y):
                if layer + 1 < cfg.num_layers_rnn and acts['layer_in'][layer + 1] is y and \
                        acts['in_split'][layer + 1] is not None:
                    return acts['in_split'][layer + 1]
                if layer + 1 == cfg.num_layers_rnn and acts['flat_of'] is y and \
                        acts['flat_split'] is not None:
                    return acts['flat_split']
                if acts['out_split'][layer] is None:
                    acts['out_split'][layer] = split_gemm.split(y.view(rows, 2 * hidden),
                                                                split_gemm.A_ORDER)
                return acts['out_split'][layer]

            side_tensors = [dxw] + [t.buf for t in (ds, drs, acts['in_split'][i],
                                                    acts['flat_split']) if t is not None] + \
                [t[0].buf for t in (x16, y16)
                 if t is not None and not isinstance(t, split_gemm.ColScaled)]

            def split_steps(lo, hi, ds=ds, drs=drs, dxw2d=dxw2d, drec=drec):
                # pieces of dxw (GRU: and drec) for steps [lo, hi) of both directions
                for d, (a, b) in enumerate(((lo, hi), (t_out - hi, t_out - lo))):
                    rng, cols = slice(a * batch, b * batch), slice(d * gh, (d + 1) * gh)
                    hip.split_bf16(dxw2d[rng, cols], split_gemm.B_ORDER, out=ds.buf[rng, :, cols])
                    if drs is not ds:
                        hip.split_bf16(drec.view(rows, 2 * gh)[rng, cols], split_gemm.B_ORDER,
                                       out=drs.buf[rng, :, cols])

            x_packs = {}       # (first row, rows) -> the layer input's rows packed for the own kernel

            def partial_weight_grads_f16(lo, hi, name=name, dxw2d=dxw2d, drec=drec, x16=x16,
                                         y16=y16, x=x, y=y, colmax=None, x_packs=x_packs):
                # steps [lo, hi) in the fp16 form: per direction one column-scaled split of the
                # finished rows of dxw (GRU: and of drec) feeds both W_ih's and W_hh's product
                # (``colmax``: the column maxima of these rows, left by the recurrence launch)
                own = (self.own_wgrad and cell != 'gru' and
                       not isinstance(x16, split_gemm.ColScaled) and
                       not isinstance(y16, split_gemm.ColScaled))
                for d, (a, b) in enumerate(((lo, hi), (t_out - hi, t_out - lo))):
                    cols = slice(d * gh, (d + 1) * gh)
                    if own:
                        # one launch for both matrices: dxw's rows packed transposed with column
                        # scales, the input's rows and the output's rows one step back / ahead
                        # (zeros outside the sequence: h(-1) = h(T) = 0) with their fixed scales
                        n_rows = (b - a) * batch
                        stages = (n_rows + 31) // 32
                        d_rows = dxw2d[a * batch:b * batch, cols]
                        scale, inv = hip.colmax_scale(d_rows) if colmax is None else \
                            hip.colscale_from_max(colmax[d * gh:(d + 1) * gh])
                        d_pk = hip.wgrad16_pack(d_rows, n_rows, 0, stages, 1.0, col_scale=scale)
                        # (the two directions meet the same rows of the layer's input in
                        # different launches - times [lo, hi) here, mirrored there: packed once)
                        x_pk = x_packs.pop((a, n_rows), None)
                        if x_pk is None:
                            x_pk = hip.wgrad16_pack(x.view(rows, -1), rows, a * batch, stages,
                                                    x16[1])
                            x_packs[(a, n_rows)] = x_pk
                        y_pk = hip.wgrad16_pack(
                            y.view(rows, 2 * hidden)[:, d * hidden:(d + 1) * hidden], rows,
                            (a + (1 if d else -1)) * batch, stages, y16[1])
                        hip.wgrad16_gemm(d_pk, gh, stages, inv, x_pk, 0, x16[1],
                                         g[name + '/w_ih'][d], y_packed=y_pk, y_scale=y16[1],
                                         dw_y=g[name + '/w_hh'][d], parts=self.own_wgrad_parts)
                        continue
                    d16, inv = split_gemm.wgrad16_operand(
                        dxw2d[a * batch:b * batch, cols],
                        None if colmax is None else colmax[d * gh:(d + 1) * gh])
                    split_gemm.wgrad16(g[name + '/w_ih'][d], d16, inv, x16[0], x16[1], a * batch,
                                       x_col_inv=getattr(x16, 'col_inv', None))
                    if d == 0:
                        a2, b2, shift, hcols = max(a, 1), b, -1, slice(0, hidden)
                    else:
                        a2, b2, shift, hcols = a, min(b, t_out - 1), 1, slice(hidden, 2 * hidden)
                    if b2 <= a2:
                        continue
                    if cell == 'gru':
                        d16, inv = split_gemm.wgrad16_operand(
                            drec.view(rows, 2 * gh)[a * batch:b * batch, cols])
                    split_gemm.wgrad16(g[name + '/w_hh'][d], d16, inv, y16[0], y16[1],
                                       (a2 + shift) * batch, x_cols=hcols,
                                       d_rows=slice((a2 - a) * batch, (b2 - a) * batch),
                                       x_col_inv=getattr(y16, 'col_inv', None))

            def partial_weight_grads(lo, hi, name=name, x=x, y=y, dxw=dxw, drec=drec,
                                     ds=ds if self.split_wgrad else None, drs=drs,
                                     input_pieces=input_pieces, output_pieces=output_pieces,
                                     g16=g16, f16_form=partial_weight_grads_f16, colmax=None):
                if g16:
                    return f16_form(lo, hi, colmax=colmax)
                # steps [lo, hi): times [lo, hi) of the forward direction, mirrored for the other
                x3 = x.view(t_out, batch, -1)
                xs = input_pieces() if ds is not None else None
                ys = output_pieces() if ds is not None else None
                for d, (a, b) in enumerate(((lo, hi), (t_out - hi, t_out - lo))):
                    cols = slice(d * gh, (d + 1) * gh)
                    if ds is not None:
                        split_gemm.mm_tn_rows(g[name + '/w_ih'][d], ds, xs, a * batch, b * batch,
                                              a_cols=cols)
                    else:
                        g[name + '/w_ih'][d].addmm_(
                            dxw[a:b, :, d, :].reshape((b - a) * batch, gh).t(),
                            x3[a:b].reshape((b - a) * batch, -1))
                    # dW_hh[d] += drec_t^T h_(t-1) (forward) / h_(t+1) (backward direction)
                    if d == 0:
                        a, shift, hcols = max(a, 1), -1, slice(0, hidden)
                    else:
                        b, shift, hcols = min(b, t_out - 1), 1, slice(hidden, 2 * hidden)
                    if b <= a:
                        continue
                    if ds is not None:
                        split_gemm.mm_tn_rows(g[name + '/w_hh'][d], drs, ys, a * batch, b * batch,
                                              a_cols=cols, b_cols=hcols, b_shift=shift * batch)
                    else:
                        g[name + '/w_hh'][d].addmm_(
                            drec[a:b, :, d, :].reshape((b - a) * batch, gh).t(),
                            y[a + shift:b + shift, :, hcols].reshape((b - a) * batch, hidden))

            # the bias gradients - column sums of dxw (GRU: and of drec) - come out of the
            # recurrence kernels themselves (accumulated into the zeroed arena slices: b_ih, and
            # for the GRU b_hh right behind it), not out of extra passes over dxw
            b_start = self.arena.offsets[name + '/b_ih']
            b_count = 2 * gh * (2 if cell == 'gru' else 1)
            assert self.arena.offsets[name + '/b_hh'] == b_start + 2 * gh
            dbias = self.arena.grad[b_start:b_start + b_count]
            bounds = [t_out * (chunks - c) // chunks for c in range(chunks + 1)]   # T ... 0
            split_done = []
            # dgates W_hh on the fp16 matrix pipe where the kernel exists (LSTM-1024); it then also
            # leaves the column maxima of each launch's rows of dxw for the fp16 weight gradients
            bwd_flags = self.rnn_bwd_flags | ((hip.RNN_F16 | self.rnn_xcd_flag |
                                               self.rnn_stagger_flag |
                                               (hip.RNN_KPAIR if hidden == 2048 and
                                                self.rnn_kpair_wide else 0))
                                              if self.rnn_bwd_f16 else 0)
            f16_rec = hip.rnn_f16_recurrence(cell, t_out, batch, hidden, bwd_flags, backward=True,
                                             ragged=acts['rnn_len'] is not None)
            arith['rnn{}/recurrence_bwd'.format(i)] = 'fp16x3' if f16_rec else 'fp32'
            colmax = torch.zeros((chunks, 2 * gh), dtype=torch.int32, device=dy.device) \
                if f16_rec and g16 else None
            if colmax is not None:
                side_tensors.append(colmax)
            need_dx = i > 0 or need_dx_first
            own_dg = (need_dx and g16 and w_pieces.dg16 is not None and f16_rec and
                      acts['rnn_len'] is None and
                      hip.dgrad16_supported(cell, t_out, batch, hidden))
            early_dg = own_dg and chunks > 1 and self.dgrad_early in ('side', 'own') and \
                side is not main
            dx2d, dg_done = None, []
            if early_dg:
                # every share accumulates (odd T': the ranges of the two directions overlap by a row)
                dx2d = torch.zeros((rows, w_ih.shape[1]), dtype=torch.float32, device=dy.device)
                side_tensors.append(dx2d)

            def dgrad_share(lo, hi, dx2d=dx2d, w_pieces=w_pieces, n=w_ih.shape[1]):
                # steps [lo, hi) of the pass: direction 0's times [lo, hi), direction 1's mirrored
                for d, (a, b) in enumerate(((lo, hi), (t_out - hi, t_out - lo))):
                    hip.dgrad16_blockscaled(acts['rnn_ws'], t_out, batch, hidden, w_pieces.dg16,
                                            split_gemm.W_SCALE, n, out=dx2d, steps=(a, b),
                                            dirs=(d, d + 1), accumulate=True)

            for c in range(chunks):
                hip.rnn_bwd(cell, dy, y, self._w_hh_t[i], acts['reserves'][i], acts['rnn_len'],
                            dxw=dxw, dbias=dbias, workspace=acts['rnn_ws'],
                            steps=(bounds[c + 1], bounds[c]), flags=bwd_flags,
                            ticket=self._take_ticket() if persistent and not whole_chip_rnn
                            else 0, colmax=None if colmax is None else colmax[c])
                if c + 1 < chunks:
                    if early_dg and self.dgrad_early == 'own':
                        if self._dgrad_stream is None:
                            self._dgrad_stream = torch.cuda.Stream(self.device)
                        ready = torch.cuda.Event()
                        ready.record(main)
                        with torch.cuda.stream(self._dgrad_stream):
                            self._dgrad_stream.wait_event(ready)
                            if persistent:      # (let the next launch take its 128 CUs first)
                                self._gate_side_stream(cell, acts['rnn_ws'], t_out, batch, hidden)
                            dgrad_share(bounds[c + 1], bounds[c])
                            dg_done.append(torch.cuda.Event())
                            dg_done[-1].record(self._dgrad_stream)
                        dx2d.record_stream(self._dgrad_stream)

                    def finished_steps(lo=bounds[c + 1], hi=bounds[c],
                                       colmax=None if colmax is None else colmax[c]):
                        if early_dg and self.dgrad_early == 'side':
                            dgrad_share(lo, hi)
                            dg_done.append(torch.cuda.Event())
                            dg_done[-1].record(torch.cuda.current_stream(self.device))
                        if ds is not None:  # these steps' pieces: beside the next launch as well
                            split_steps(lo, hi)
                            split_done.append(torch.cuda.Event())
                            split_done[-1].record(torch.cuda.current_stream(self.device))
                        partial_weight_grads(lo, hi, colmax=colmax)
                    on_side(side_tensors, finished_steps, gate=True)
            if ds is not None:
                split_steps(0, bounds[-2])
                for event in split_done:
                    main.wait_event(event)
            # critical path: the gradient w.r.t. this layer's input feeds the layer below.
            # ONE library GEMM at a time: the weight-gradient GEMMs still queued on the side
            # stream finish first.  Two stream-K GEMMs of the library in flight at once - each
            # holding CUs while it waits for partial tiles of workgroups that have no CU yet - can
            # wait for each other forever (seen at C2: the bf16 data gradient [8000 x 640] on this
            # stream against a [4096 x 640] weight gradient on the side stream, second training
            # step); GEMMs beside this package's own kernels are fine, those never wait on them.
            dy_below = None
            if need_dx:
                arith['rnn{}/data_gradient'.format(i)] = \
                    'fp16x3 block-scaled (own kernel)' if own_dg else \
                    'fp16x3' if g16 else 'bf16x6' if use_split else 'fp32'
                if early_dg:
                    # the earlier launches' shares ran beside the launches after them; the last
                    # launch's share here
                    for event in dg_done:
                        main.wait_event(event)
                    dgrad_share(0, bounds[-2])
                    dy_below = dx2d.view(t_out, batch, -1)
                elif own_dg:
                    # straight from the pieces the recurrence launches above published into the
                    # workspace (valid until the next persistent launch): an own kernel that
                    # waits for no other workgroup - the side stream keeps its backlog
                    dy_below = hip.dgrad16_blockscaled(
                        acts['rnn_ws'], t_out, batch, hidden, w_pieces.dg16, split_gemm.W_SCALE,
                        w_ih.shape[1]).view(t_out, batch, -1)
                elif g16:
                    if w_pieces[3] is None:     # (packed for the own kernel, which did not apply)
                        w_pieces.make_tr16()
                    # (the row split of dxw runs beside what is left of the side stream's
                    # backlog - 0.2 - 0.3 ms per C3 layer; only the GEMM has to wait for it)
                    dy_below = split_gemm.dgrad16(
                        dxw2d, w_pieces[3], split_gemm.W_SCALE,
                        before_gemm=(lambda: main.wait_stream(side)) if side is not main
                        else None).view(t_out, batch, -1)
                elif side is not main:
                    main.wait_stream(side)
                if g16:
                    pass                    # (done above)
                elif use_split:
                    dy_below = torch.empty((rows, x.shape[-1]), dtype=torch.float32,
                                           device=dy.device)
                    split_gemm.mm_nt_by_order(dy_below, ds, w_pieces[1])
                    dy_below = dy_below.view(t_out, batch, -1)
                else:
                    dy_below = torch.mm(dxw2d, w_ih).view(t_out, batch, -1)
                if seeds[0] is not None:
                    dy_below = hip.dropout(dy_below, rnn_rate, seeds[0])

            def weight_grads(name=name, x=x, y=y, dxw=dxw, dxw2d=dxw2d, i=i, chunks=chunks,
                             last=bounds[-2], partial_weight_grads=partial_weight_grads,
                             drec=drec, use_split=use_split,
                             colmax=None if colmax is None else colmax[chunks - 1]):
                if cell != 'gru':       # (the GRU's db_hh came out of the kernel with db_ih)
                    g[name + '/b_hh'].copy_(g[name + '/b_ih'])
                if chunks > 1 or use_split:   # the earlier launches' shares are already in
                    partial_weight_grads(0, last, colmax=colmax)
                    return
                torch.mm(dxw2d.t(), x.view(rows, -1),
                         out=g[name + '/w_ih'].view(2 * gates * hidden, -1))
                # (drec: the gradient w.r.t. the recurrent pre-activations - dxw itself, except for
                # the GRU)
                # dW_hh[d] = sum_t drec_t^T h_{t-1}: one GEMM per direction over shifted views
                if t_out > 1:
                    gh = gates * hidden
                    torch.mm(drec[1:, :, 0, :].reshape((t_out - 1) * batch, gh).t(),
                             y[:-1, :, :hidden].reshape((t_out - 1) * batch, hidden),
                             out=g[name + '/w_hh'][0])
                    torch.mm(drec[:-1, :, 1, :].reshape((t_out - 1) * batch, gh).t(),
                             y[1:, :, hidden:].reshape((t_out - 1) * batch, hidden),
                             out=g[name + '/w_hh'][1])

            def layer_weight_grads(name=name, weight_grads=weight_grads):
                weight_grads()
                if early:
                    done(name)

            on_side(side_tensors, layer_weight_grads, gate=i > 0, beside_recurrence=i > 0)
            if not early:
                deferred.append(name)
            if dy_below is not None:
                dy = dy_below
        # The deferred layers' gradients are final once the side stream has drained.  Their hooks
        # run with the side stream current, so a bucketed all-reduce launched from them is ordered
        # after the weight-gradient GEMMs without the main stream having to wait: the front-end
        # backward below overlaps the side stream's tail.
        with torch.cuda.stream(side):
            for name in deferred:
                done(name)
            if hasattr(reduce_hook, 'flush'):
                reduce_hook.flush()     # pending slices were produced on THIS stream

        # front-end
        if cfg.used_model == 'ds2':
            conv_out = acts['conv_out']
            last = conv_out[-1]
            if acts['last_time_major']:
                # the last layer's output - and so its gradient - is time-major [T', B, F', C]
                dact = dy.contiguous().view(last.shape)
            else:
                b_, c_, tt, ff = last.shape
                # [T', B, F'*C] -> NHWC storage [B, T', F', C] for the gradient as well
                dact = dy.view(tt, b_, ff, c_).permute(1, 0, 2, 3).contiguous()
            if side is not main and any(kind is None for kind in acts['conv_own']):
                main.wait_stream(side)      # MIOpen may run library GEMMs: one at a time
            for i in range(len(cfg.conv_filters) - 1, -1, -1):
                name = 'conv{}'.format(i)
                tm = acts['last_time_major'] and i == len(cfg.conv_filters) - 1
                out_phys = conv_out[i] if tm else conv_out[i].permute(0, 2, 3, 1)
                own = acts['conv_own'][i]
                # Own kernels, no conv dropout: the epilogue's backward pass - the mask of
                # min(max(., 0), cutoff) and the bias gradient - happens inside the gradient
                # kernels while they stage dz (round 3; before: an elementwise pass per layer)
                fused_bwd = (self.conv_fused_bwd and own is not None and
                             cfg.conv_dropout_rate == 0.0)
                if fused_bwd:
                    dz = dact if tm else dact.permute(0, 3, 1, 2)
                    mask = dict(act=out_phys, relu_cutoff=cfg.relu_cutoff)
                else:
                    # the bias gradient (sum of dz over batch, time and frequency) falls out of
                    # the epilogue's backward pass: channels are the columns of the [.., F, C] view
                    dz = hip.bias_act_bwd(out_phys, dact, cfg.relu_cutoff,
                                          cfg.conv_dropout_rate, g[name + '/bias'])
                    if not tm:
                        dz = dz.permute(0, 3, 1, 2)    # logical NCHW view of the NHWC storage
                    mask = {}
                pt0, pt1, pf0, pf1 = acts['pads'][i]
                if own == 'conv0':
                    # first layer: kernel gradient straight from the features, no padded copy
                    wrw0 = hip.conv0_wrw16 if (self.conv_f16 and self.conv_wrw_f16) \
                        else hip.conv0_wrw
                    wrw0(dz.permute(0, 2, 3, 1), acts['features'], out=g[name + '/kernel'],
                         dbias=g[name + '/bias'] if fused_bwd else None, **mask)
                    done(name)
                    continue
                conv_in = acts['conv_in'][i]
                if own == 's12':
                    # own kernels: kernel gradient straight from the unpadded NHWC input
                    # (deterministic two-stage reduction), then the data gradient (weights were
                    # packed by the forward pass of this step)
                    dz_phys = dz if tm else dz.permute(0, 2, 3, 1)
                    if acts['conv_f16'].get(i) and self.conv_wrw_f16:
                        # (the layer's input is bounded: the scale the forward pass used)
                        hip.conv_s12_wrw16(dz_phys, conv_in.permute(0, 2, 3, 1),
                                           acts['conv_f16'][i], out=g[name + '/kernel'],
                                           time_major=tm,
                                           dbias=g[name + '/bias'] if fused_bwd else None, **mask)
                    else:
                        hip.conv_s12_wrw(dz_phys, conv_in.permute(0, 2, 3, 1),
                                         out=g[name + '/kernel'], time_major=tm,
                                         dbias=g[name + '/bias'] if fused_bwd else None, **mask)
                    if i > 0 and acts['conv_f16'].get(i):
                        # (the fp16 pieces of this step's weights, packed by the forward pass)
                        dact = hip.conv_s12_bwd_data16(dz_phys, self._conv_packed16[i],
                                                       time_major=tm, **mask)
                    elif i > 0:
                        dact = hip.conv_s12_bwd_data(dz_phys, self._conv_packed[i],
                                                     time_major=tm, **mask)
                    done(name)
                    continue
                need_dx = i > 0
                dxp, dw, _ = torch.ops.aten.convolution_backward(
                    dz, conv_in, self._conv_kernel_cl(i), [p[name + '/bias'].shape[0]],
                    list(CONV_STRIDES[i]), [0, 0], [1, 1], False, [0, 0], 1,
                    [need_dx, True, False])
                g[name + '/kernel'].copy_(dw)
                if need_dx:
                    dact = dxp[:, :, pt0:dxp.shape[2] - pt1, pf0:dxp.shape[3] - pf1] \
                        .permute(0, 2, 3, 1).contiguous()
                done(name)
        else:
            if side is not main:
                main.wait_stream(side)      # one library GEMM at a time (see above)
            dact = dy.reshape(rows, -1)
            for i in range(2, -1, -1):
                name = 'dense{}'.format(i)
                dz = hip.bias_act_bwd(acts['dense_out'][i], dact.contiguous(), cfg.relu_cutoff,
                                      cfg.dense_dropout_rate if training else 0.0,
                                      g[name + '/bias'])
                torch.mm(acts['dense_in'][i].t(), dz, out=g[name + '/kernel'])
                if i > 0:
                    dact = torch.mm(dz, p[name + '/kernel'].t())
                done(name)
        if side is not main:
            main.wait_stream(side)

    def forward_backward(self, features, feature_len, labels, reduce_hook=None, check=True):
        """One training forward + backward; returns the mean CTC loss (device scalar)."""
        logits, seq_length = self.inference_fn(features, feature_len, training=True)
        loss = self.loss_fn(logits, seq_length, labels, check=check)
        self.backward(reduce_hook)
        return loss

    def step_guard(self):
        """int32[2] device tensor for `apply_gradients(skip=...)`: [0] != 0 when the gradients
        of the last `forward_backward` must not be applied - a CTC status word set (infeasible
        alignment / bad labels: where ``tf.nn.ctc_loss`` raises), a non-finite loss, or a
        persistent recurrence kernel that gave up at a barrier - decided on the device, so the
        host may find out later (deferred checks) without the parameters having been touched;
        [1] = the recurrence time-out words (`engine.Trainer` copies this to pinned memory)."""
        acts, cfg = self._acts, self.cfg
        words = (0, 0)
        if acts is not None and hip.rnn_persistent_supported(cfg.cell, acts['t_out'],
                                                             acts['batch'], cfg.num_units_rnn):
            words = hip.rnn_timeout_words(cfg.cell, acts['rnn_ws'], acts['t_out'], acts['batch'],
                                          cfg.num_units_rnn)
        return hip.step_guard(self.last_status, self.last_per_utterance_loss, words)

    def apply_gradients(self, learning_rate=1e-5, beta1=0.9, beta2=0.999, epsilon=1e-8,
                        grad_scale=1.0, skip=None):
        """TensorFlow-form Adam over the whole arena in one launch (``asr/model.py:80-83``).
        ``skip``: the device flag of `step_guard` - parameters and moments stay untouched when
        it is set (the step counter still advances)."""
        self.step_count += 1
        a = self.arena
        hip.adam_step(a.param, a.grad, a.m, a.v, self.step_count, learning_rate, beta1, beta2,
                      epsilon, grad_scale, skip=skip)

    # ------------------------------------------------------------------ estimator-style entry
    def model_fn(self, features, labels, mode, learning_rate=1e-5, adam=(0.9, 0.999, 1e-8)):
        """One call of the reference's ``model_fn`` (``asr/model.py:23-121``) executed eagerly.

        ``mode``: 'infer' (tf.estimator.ModeKeys.PREDICT) -> {'decoded', 'plaintext'};
        'train' -> runs forward, loss, backward and one Adam step, returns {'loss', 'decoded',
        'plaintext', 'mean_edit_distance', 'word_error_rate'}; 'eval' -> the same without the
        update.  Anything else raises ``RuntimeError('Invalid mode.')`` like the reference."""
        if mode not in ('train', 'eval', 'infer'):
            raise RuntimeError('Invalid mode.')
        logits, seq_length = self.inference_fn(features['spectrogram'],
                                               features['spectrogram_length'],
                                               training=(mode == 'train'))
        self.check_rnn_error()
        if mode == 'infer':
            decoded, plaintext, _ = self.decode_fn(logits, seq_length, None)
            return {'decoded': decoded, 'plaintext': plaintext}
        loss = self.loss_fn(logits, seq_length, labels)
        if mode == 'train':
            self.backward()
            self.apply_gradients(learning_rate, *adam)
        originals = features['label_plaintext']
        decoded, plaintext, summary = self.decode_fn(
            logits, seq_length, np.array([t.encode('utf-8') if isinstance(t, str) else t
                                          for t in originals], dtype=object))
        _, mean_ed, _, wer = self.error_rates_fn(labels, [t.decode('utf-8') if isinstance(t, bytes)
                                                          else t for t in originals],
                                                 decoded, plaintext)
        return {'loss': loss, 'decoded': decoded, 'plaintext': plaintext, 'summary': summary,
                'mean_edit_distance': mean_ed, 'word_error_rate': wer}

    # ------------------------------------------------------------------ decode / score
    def decode_fn(self, logits, seq_len, originals=None, beam_width=None, greedy=False):
        """CTC decode + plaintext (``asr/model.py:271-309``).  Returns (decoded: list of B int
        lists — the values of the reference's SparseTensor —, plaintext object[B], summary
        object[2, B]).  ``greedy=True`` selects the greedy decoder instead of the beam search."""
        beam_width = self.cfg.beam_width if beam_width is None else beam_width
        if greedy:
            out, out_len = hip.ctc_greedy_decode(logits, seq_len)
        else:
            out, out_len, _ = hip.ctc_beam_decode(logits, seq_len, beam_width)
        return self._decoded_to_text(out.cpu().numpy(), out_len.cpu().numpy(), originals)

    @staticmethod
    def _decoded_to_text(out, out_len, originals):
        decoded = [out[b, :out_len[b]].tolist() for b in range(out.shape[0])]
        width = max([len(d) for d in decoded] + [0])
        dense = np.zeros((len(decoded), width), dtype=np.int32)
        for b, row in enumerate(decoded):
            dense[b, :len(row)] = row
        plaintext, summary = metrics.dense_to_text(dense, originals if originals is not None
                                                   else np.array([], dtype=np.int32))
        return decoded, plaintext, summary

    def decode_group_size(self, num_steps, batch, beam_width=None, budget_bytes=48 << 30,
                          max_utterances=256):
        """How many batches of ``batch`` utterances `decode_many` should be given at once: the
        beam search runs one workgroup per utterance, so a single batch occupies 16 or 32 of the
        256 CUs and a group of batches decodes in about the time of one.  Bounded by the prefix
        tree pool (256 MB per utterance at width 1024, T' = 500) and ``max_utterances``."""
        beam_width = self.cfg.beam_width if beam_width is None else beam_width
        per_utt = max(1, hip.ctc_beam_workspace_bytes(num_steps, 1, self.cfg.num_classes,
                                                      beam_width))
        utterances = max(batch, min(max_utterances, budget_bytes // per_utt))
        return max(1, int(utterances // batch))

    def decode_many(self, batches, beam_width=None):
        """Beam-search decode of several batches in ONE launch.  ``batches`` is a list of
        (logits [T'_i, B_i, C], seq_len [B_i], originals or None); returns one `decode_fn` result
        per batch - identical to decoding the batches one by one (an utterance's search does not
        depend on its neighbours; frames past its length are never read)."""
        beam_width = self.cfg.beam_width if beam_width is None else beam_width
        if not batches:
            return []
        steps = max(int(logits.shape[0]) for logits, _, _ in batches)
        total = sum(int(logits.shape[1]) for logits, _, _ in batches)
        classes = int(batches[0][0].shape[2])
        joint = torch.zeros((steps, total, classes), dtype=torch.float32, device=self.device)
        lengths = torch.empty(total, dtype=torch.int32, device=self.device)
        start = 0
        for logits, seq_len, _ in batches:
            stop = start + int(logits.shape[1])
            joint[:logits.shape[0], start:stop] = logits
            lengths[start:stop] = seq_len
            start = stop
        out, out_len, _ = hip.ctc_beam_decode(joint, lengths, beam_width)
        out, out_len = out.cpu().numpy(), out_len.cpu().numpy()
        results, start = [], 0
        for logits, _, originals in batches:
            stop = start + int(logits.shape[1])
            results.append(self._decoded_to_text(out[start:stop], out_len[start:stop], originals))
            start = stop
        return results

    @staticmethod
    def error_rates_fn(labels, originals, decoded, decoded_texts):
        """(edit distances f32[B], mean edit distance, WERs f32[B], mean WER)
        (``asr/model.py:311-345``).  ``labels`` dense zero-padded ints or list of lists."""
        if isinstance(labels, torch.Tensor):
            labels = labels.cpu().numpy()
        truths = [[int(v) for v in row if int(v) != 0] for row in labels]
        edit_distances, mean_ed = metrics.edit_distance_batch(decoded, truths)
        wers, wer = metrics.wer_batch(list(originals), list(decoded_texts))
        return edit_distances, mean_ed, wers, wer
