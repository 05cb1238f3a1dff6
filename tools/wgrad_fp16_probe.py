#!/usr/bin/env python
"""Would the WEIGHT-gradient GEMMs dW = d^T x tolerate two fp16 pieces with a power-of-two scale
per column of d (per gate unit; x is bounded), three products?  Takes the real operands of a
training step (C3-like stack, real CTC gradients) by capturing what `split_gemm.mm_tn_rows` is
given, and compares against fp64: library fp32 GEMM, bf16 x 6, fp16 x 3 with per-column scales.
    python tools/wgrad_fp16_probe.py"""
import json
import os
import sys

os.environ['CTCASR_BWD_F16'] = '0'      # the capture below hooks the bf16 form's weight-gradient call
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from ctc_asr_amd import hip, split_gemm as sg
from ctc_asr_amd.model import CTCModel, ModelConfig

hip.load()
captured = []
orig = sg.mm_tn_rows


def spy(out, a, b, lo, hi, a_cols=slice(None), b_cols=slice(None), b_shift=0, accumulate=True):
    if len(captured) < 40:
        captured.append((a, b, lo, hi, a_cols, b_cols, b_shift))
    return orig(out, a, b, lo, hi, a_cols, b_cols, b_shift, accumulate)


sg.mm_tn_rows = spy
cfg = ModelConfig(used_model='ds2', conv_filters=(32, 32), num_units_dense=2048, num_layers_rnn=3,
                  num_units_rnn=1024, rnn_cell='lstm', cudnn=True, dense_dropout_rate=0.1)
model = CTCModel(cfg, 'cuda', seed=3)
rng = np.random.default_rng(0)
batch, frames = 32, 999
feats = torch.tensor(rng.normal(size=(batch, frames, 80)).astype(np.float32), device='cuda')
flen = torch.full((batch,), frames, dtype=torch.int32)
labels = [list(rng.integers(1, 28, size=120)) for _ in range(batch)]
for _ in range(3):                         # a few steps so that the weights are not the initialiser's
    model.forward_backward(feats, flen, labels)
    model.apply_gradients(1e-3)
captured.clear()
model.forward_backward(feats, flen, labels)
torch.cuda.synchronize()


def pieces_sum(s):          # fp32 operand back from its three bf16 pieces
    return s.piece(0).float() + s.piece(1).float() + s.piece(2).float()


def err(got, ref):
    d = got.double() - ref
    return [float(d.pow(2).mean().sqrt() / ref.pow(2).mean().sqrt().clamp_min(1e-300)),
            float(d.abs().max() / ref.abs().max().clamp_min(1e-300))]


def split_fp16_cols(a, scale_cols):
    s = a * scale_cols
    a1 = s.to(torch.float16)
    return a1, (s - a1.float()).to(torch.float16)


seen = set()
for a, b, lo, hi, a_cols, b_cols, b_shift in captured:
    key = (a.rows, a.cols, b.cols, lo, hi, a_cols.start, b_cols.start, b_shift)
    if key in seen or len(seen) >= 8:
        continue
    seen.add(key)
    # d: rows [lo, hi) x a_cols of operand a; x: rows shifted x b_cols of operand b
    first_is_grad = a.order == sg.B_ORDER           # (dense4 passes the activations first)
    A = pieces_sum(a)[lo:hi, a_cols]
    B = pieces_sum(b)[lo + b_shift:hi + b_shift, b_cols]
    ref = A.double().t() @ B.double()
    row = {'shape': [A.shape[1], A.shape[0], B.shape[1]],
           'operand range': [float(A.abs().max()), float(A.abs()[A != 0].min()),
                             float(B.abs().max())]}
    row['fp32 GEMM'] = err(torch.mm(A.t(), B), ref)
    out = torch.zeros(A.shape[1], B.shape[1], device='cuda')
    orig(out, a, b, lo, hi, a_cols, b_cols, b_shift, False)
    row['bf16 x 6'] = err(out, ref)
    # fp16 x 3: per-column power-of-two scales from the column maxima of BOTH operands
    res = None
    for name, per_col_b in (('fp16 x 3, per-column scales for both', True),
                            ('fp16 x 3, per-column scale for the first, one scale for the second', False)):
        sa = torch.exp2(14.0 - torch.ceil(torch.log2(A.abs().amax(dim=0).clamp_min(1e-30))))
        if per_col_b:
            sb = torch.exp2(14.0 - torch.ceil(torch.log2(B.abs().amax(dim=0).clamp_min(1e-30))))
        else:
            sb = torch.exp2(14.0 - torch.ceil(torch.log2(B.abs().max().clamp_min(1e-30)))) * \
                torch.ones(B.shape[1], device='cuda')
        a1, a2 = split_fp16_cols(A, sa)
        b1, b2 = split_fp16_cols(B, sb)
        ka = torch.cat([a1, a1, a2], dim=0)
        kb = torch.cat([b1, b2, b1], dim=0)
        res = torch.mm(ka.t(), kb, out_dtype=torch.float32) / (sa[:, None] * sb[None, :])
        row[name] = err(res, ref)
    print(json.dumps(row), flush=True)
