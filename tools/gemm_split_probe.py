#!/usr/bin/env python
"""Can the fp32 projections of the C3 step run on the bf16 matrix pipe without giving up fp32
accuracy?  gfx950 has no xf32 and its fp32 MFMA runs at 1/16 of the bf16 rate, so an fp32 operand
split into three bf16 pieces (a = a1 + a2 + a3, 24 mantissa bits in all) and multiplied as the six
products of order <= 2 (a1b1; a1b2 + a2b1; a1b3 + a2b2 + a3b1) with fp32 accumulation costs
6 / 16 of the fp32-MFMA time on paper.  This probe measures what hipBLASLt makes of it for the
three GEMM shapes of a C3 layer and what the result's error against fp64 is next to the plain
fp32 GEMM's.

    python tools/gemm_split_probe.py [--reps 40]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def split3(a):
    import torch
    a1 = a.to(torch.bfloat16)
    r = a - a1.float()
    a2 = r.to(torch.bfloat16)
    r = r - a2.float()
    return a1, a2, r.to(torch.bfloat16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--reps', type=int, default=40)
    ap.add_argument('--rows', type=int, default=16000)
    args = ap.parse_args()
    import torch
    torch.manual_seed(0)
    dev = 'cuda'
    R, F, G = args.rows, 2048, 8192
    report = {}

    def timed(fn, reps=args.reps, warm=12):
        for _ in range(warm):
            fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    def errors(got, ref):
        d = (got.double() - ref)
        return {'max_abs': float(d.abs().max()), 'rms_rel': float(d.pow(2).mean().sqrt() /
                                                                  ref.pow(2).mean().sqrt())}

    x = torch.randn(R, F, device=dev).clamp_(0, 20) * (torch.rand(R, F, device=dev) < 0.5)
    w = torch.randn(G, F, device=dev) / F ** 0.5
    d = torch.randn(R, G, device=dev) * 1e-3 * torch.rand(R, 1, device=dev)
    flops = 2.0 * R * F * G

    # ---- forward projection  xw[R, G] = x[R, F] . w[G, F]^T  (K = F) -------------------------
    x1, x2, x3 = split3(x)
    w1, w2, w3 = split3(w)
    xa = torch.cat([x1, x2, x3, x1, x2, x1], dim=1).contiguous()          # [R, 6F]
    wb = torch.cat([w3, w2, w1, w2, w1, w1], dim=1).contiguous()          # [G, 6F]
    xa3 = torch.cat([x1, x1, x2], dim=1).contiguous()
    wb3 = torch.cat([w1, w2, w1], dim=1).contiguous()
    out = torch.empty(R, G, device=dev)
    sub = slice(0, 1024)
    ref = x[sub].double() @ w.double().t()
    t_f32 = timed(lambda: torch.mm(x, w.t(), out=out))
    e_f32 = errors(torch.mm(x[sub], w.t()), ref)
    t_b6 = timed(lambda: torch.mm(xa, wb.t(), out_dtype=torch.float32))
    e_b6 = errors(torch.mm(xa[sub], wb.t(), out_dtype=torch.float32), ref)
    t_b3 = timed(lambda: torch.mm(xa3, wb3.t(), out_dtype=torch.float32))
    e_b3 = errors(torch.mm(xa3[sub], wb3.t(), out_dtype=torch.float32), ref)
    t_b1 = timed(lambda: torch.mm(x1, w1.t(), out_dtype=torch.float32))
    report['fwd'] = {
        'shape': [R, F, G],
        'fp32_ms': round(t_f32, 3), 'fp32_tflops': round(flops / t_f32 / 1e9, 1), 'fp32_err': e_f32,
        'bf16x6_ms': round(t_b6, 3), 'bf16x6_raw_tflops': round(6 * flops / t_b6 / 1e9, 1),
        'bf16x6_err': e_b6,
        'bf16x3_ms': round(t_b3, 3), 'bf16x3_raw_tflops': round(3 * flops / t_b3 / 1e9, 1),
        'bf16x3_err': e_b3,
        'bf16x1_ms': round(t_b1, 3), 'bf16x1_raw_tflops': round(flops / t_b1 / 1e9, 1)}
    print(json.dumps(report['fwd']), flush=True)

    # the split itself: one pass over x writing the six-block layout (torch ops here; a fused
    # kernel reads 4 and writes 12 bytes per element)
    def do_split():
        a1, a2, a3 = split3(x)
        torch.cat([a1, a2, a3, a1, a2, a1], dim=1, out=xa)
    report['split_x_torch_ms'] = round(timed(do_split, reps=10, warm=3), 3)
    del xa, wb, xa3, wb3, out

    # ---- weight gradient  dw[G, F] = d[R, G]^T . x[R, F]  (K = R), three calls, blocked pieces -
    d1, d2, d3 = split3(d)
    db = torch.cat([d1, d2, d3], dim=0).contiguous()                       # [3R, G]
    xb = torch.cat([x3, x2, x1], dim=0).contiguous()                       # [3R, F]
    ref = d.double().t()[sub] @ x.double()
    dw = torch.empty(G, F, device=dev)
    t_f32 = timed(lambda: torch.mm(d.t(), x, out=dw))
    e_f32 = errors(torch.mm(d.t()[sub], x), ref)

    def wgrad_split(rows=slice(None)):
        acc = torch.mm(db.t()[rows], xb, out_dtype=torch.float32)                    # order 2
        acc = torch.addmm(acc, db[:2 * R].t()[rows], xb[R:], out_dtype=torch.float32)   # order 1
        return torch.addmm(acc, db[:R].t()[rows], xb[2 * R:], out_dtype=torch.float32)  # order 0
    t_b6 = timed(wgrad_split)
    e_b6 = errors(wgrad_split(sub), ref)
    report['wgrad'] = {'shape': [G, R, F], 'fp32_ms': round(t_f32, 3),
                       'fp32_tflops': round(flops / t_f32 / 1e9, 1), 'fp32_err': e_f32,
                       'bf16x6_ms': round(t_b6, 3),
                       'bf16x6_raw_tflops': round(6 * flops / t_b6 / 1e9, 1), 'bf16x6_err': e_b6}
    print(json.dumps(report['wgrad']), flush=True)
    del xb

    # ---- data gradient  dx[R, F] = d[R, G] . w[G, F]  (K = G): six calls on blocked pieces ----
    wp = [w1, w2, w3]
    dp = [d1, d2, d3]
    ref = d[sub].double() @ w.double()
    dx = torch.empty(R, F, device=dev)
    t_f32 = timed(lambda: torch.mm(d, w, out=dx))
    e_f32 = errors(torch.mm(d[sub], w), ref)
    pairs = [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]             # small terms first

    def dgrad_split(rows=slice(None)):
        acc = None
        for i, j in pairs:
            if acc is None:
                acc = torch.mm(dp[i][rows], wp[j], out_dtype=torch.float32)
            else:
                acc = torch.addmm(acc, dp[i][rows], wp[j], out_dtype=torch.float32)
        return acc
    t_b6 = timed(dgrad_split)
    e_b6 = errors(dgrad_split(sub), ref)
    # one call on a six-block K-concatenation, for comparison
    da = torch.cat([d1, d2, d3, d1, d2, d1], dim=1).contiguous()          # [R, 6G]
    wk = torch.cat([w3, w2, w1, w2, w1, w1], dim=0).contiguous()          # [6G, F]
    t_b6c = timed(lambda: torch.mm(da, wk, out_dtype=torch.float32))
    e_b6c = errors(torch.mm(da[sub], wk, out_dtype=torch.float32), ref)
    report['dgrad'] = {'shape': [R, G, F], 'fp32_ms': round(t_f32, 3),
                       'fp32_tflops': round(flops / t_f32 / 1e9, 1), 'fp32_err': e_f32,
                       'bf16x6_six_calls_ms': round(t_b6, 3), 'bf16x6_six_calls_err': e_b6,
                       'bf16x6_one_call_ms': round(t_b6c, 3),
                       'bf16x6_one_call_raw_tflops': round(6 * flops / t_b6c / 1e9, 1),
                       'bf16x6_one_call_err': e_b6c}
    print(json.dumps(report['dgrad']), flush=True)
    del da, wk

    # ---- the product path (ctc_asr_amd/split_gemm.py): HIP split kernel, final layouts ----------
    from ctc_asr_amd import hip, split_gemm as sg
    hip.load()
    xs, ws, ds = sg.split(x, sg.A_ORDER), sg.split(w, sg.B_ORDER), sg.split(d, sg.B_ORDER)
    w_t = w.t().contiguous()
    wt = sg.split(w_t, sg.A_ORDER)
    assert torch.equal(xs.concat(), torch.cat([x1, x2, x3, x1, x2, x1], dim=1))
    report['split_kernel_ms'] = {
        'x [16000 x 2048] -> six blocks': round(timed(lambda: sg.split(x, sg.A_ORDER, out=xs), 20, 5), 3),
        'w [8192 x 2048] -> six blocks': round(timed(lambda: sg.split(w, sg.B_ORDER, out=ws), 20, 5), 3),
        'd [16000 x 8192] -> six blocks': round(timed(lambda: sg.split(d, sg.B_ORDER, out=ds), 20, 5), 3)}
    print(json.dumps(report['split_kernel_ms']), flush=True)
    H = F // 2
    xw = torch.empty(R, G, device=dev)
    dwh = torch.zeros(2, 4 * H, H, device=dev)
    dw2 = dw.view(2, 4 * H, F)
    chunks = [(0, R // 3), (R // 3, 2 * (R // 3)), (2 * (R // 3), R)]

    def layer_split():
        sg.split(x, sg.A_ORDER, out=xs)
        sg.mm_nt(xs, ws, out=xw)
        sg.split(d, sg.B_ORDER, out=ds)
        sg.mm_nt_by_order(dx, ds, wt)
        for lo, hi in chunks:               # weight gradients per third of the steps, as the
            for dirn in (0, 1):             # backward pass issues them
                cols = slice(dirn * 4 * H, (dirn + 1) * 4 * H)
                sg.mm_tn_rows(dw2[dirn], ds, xs, lo, hi, a_cols=cols)
                sg.mm_tn_rows(dwh[dirn], ds, xs, lo, hi, a_cols=cols,
                              b_cols=slice(dirn * H, (dirn + 1) * H))

    def layer_f32():
        torch.mm(x, w.t(), out=xw)
        torch.mm(d, w, out=dx)
        for lo, hi in chunks:
            for dirn in (0, 1):
                cols = slice(dirn * 4 * H, (dirn + 1) * 4 * H)
                dw2[dirn].addmm_(d[lo:hi, cols].t(), x[lo:hi])
                dwh[dirn].addmm_(d[lo:hi, cols].t(), x[lo:hi, dirn * H:(dirn + 1) * H])

    def layer_errors(fn):
        dw.zero_(); dwh.zero_()
        fn()
        return {'xw': errors(xw[sub], x[sub].double() @ w.double().t()),
                'dx': errors(dx[sub], d[sub].double() @ w.double()),
                'dW_ih': errors(dw[sub], d.double().t()[sub] @ x.double()),
                'dW_hh': errors(dwh[0][sub], d[:, :4 * H].double().t()[sub] @ x[:, :H].double())}

    report['layer'] = {
        'gemms': 'xw, dx, dW_ih and dW_hh (in thirds of the steps, per direction) of one C3 layer, '
                 '60 layers back to back',
        'split_err': layer_errors(layer_split), 'f32_err': layer_errors(layer_f32),
        'f32_ms': round(timed(layer_f32, 60, 10), 3),
        'split_ms': round(timed(layer_split, 60, 10), 3),
        'f32_ms_again': round(timed(layer_f32, 60, 10), 3)}
    one = {}
    for name, fn, flops_ in (
            ('forward one call K = 6 x 2048', lambda: sg.mm_nt(xs, ws, out=xw), 6 * flops),
            ('data gradient, three NT calls', lambda: sg.mm_nt_by_order(dx, ds, wt), 6 * flops),
            ('data gradient, six NN calls', lambda: sg.mm_pieces(dx, ds.piece, ws.piece), 6 * flops),
            ('W_ih gradient, one third of the steps, one direction',
             lambda: sg.mm_tn_rows(dw2[0], ds, xs, 0, R // 3, a_cols=slice(0, 4 * H)),
             6 * flops / 6)):
        ms = timed(fn, 40, 10)
        one[name] = {'ms': round(ms, 3), 'raw_bf16_tflops': round(flops_ / ms / 1e9, 0)}
    report['split_calls'] = one
    print(json.dumps(report['layer']), flush=True)
    print(json.dumps(report, indent=1))


if __name__ == '__main__':
    main()
