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

    # ---- the HIP split kernel against the torch restatement, and a whole layer's GEMMs in a loop
    from ctc_asr_amd import hip
    hip.load()
    A_ORDER, B_ORDER = (0, 1, 2, 0, 1, 0), (2, 1, 0, 1, 0, 0)
    xs = hip.split_bf16(x, A_ORDER)
    assert torch.equal(xs.view(R, 6 * F), torch.cat([x1, x2, x3, x1, x2, x1], dim=1))
    ws = hip.split_bf16(w, B_ORDER)
    ds = hip.split_bf16(d, (0, 1, 2))
    assert torch.equal(ds.view(R, 3 * G), torch.cat([d1, d2, d3], dim=1))
    report['split_kernel_ms'] = {
        'x_six_blocks': round(timed(lambda: hip.split_bf16(x, A_ORDER, out=xs), 20, 5), 3),
        'w_six_blocks': round(timed(lambda: hip.split_bf16(w, B_ORDER, out=ws), 20, 5), 3),
        'd_three_blocks': round(timed(lambda: hip.split_bf16(d, (0, 1, 2), out=ds), 20, 5), 3)}
    print(json.dumps(report['split_kernel_ms']), flush=True)
    f32 = torch.float32
    xw = torch.empty(R, G, device=dev)
    xpiece = {0: xs[:, 0], 1: xs[:, 1], 2: xs[:, 2]}                # [R, F] views, row stride 6F
    wpiece = {2: ws[:, 0], 1: ws[:, 1], 0: ws[:, 2]}
    dpiece = {0: ds[:, 0], 1: ds[:, 1], 2: ds[:, 2]}
    H = F // 2

    def layer_split():
        hip.split_bf16(x, A_ORDER, out=xs)
        torch.mm(xs.view(R, 6 * F), ws.view(G, 6 * F).t(), out_dtype=f32, out=xw)
        hip.split_bf16(d, (0, 1, 2), out=ds)
        first = True
        for i, j in pairs:
            if first:
                torch.mm(dpiece[i], wpiece[j], out_dtype=f32, out=dx)
            else:
                torch.addmm(dx, dpiece[i], wpiece[j], out_dtype=f32, out=dx)
            first = False
        first = True
        for i, j in pairs:
            if first:
                torch.mm(dpiece[i].t(), xpiece[j], out_dtype=f32, out=dw)
            else:
                torch.addmm(dw, dpiece[i].t(), xpiece[j], out_dtype=f32, out=dw)
            first = False
        for dirn in (0, 1):                 # recurrent weight gradient: [4H, R] x [R, H] per direction
            out = dwh[dirn]
            first = True
            for i, j in pairs:
                a = dpiece[i][:, dirn * 4 * H:(dirn + 1) * 4 * H].t()
                b = xpiece[j][:, dirn * H:(dirn + 1) * H]
                if first:
                    torch.mm(a, b, out_dtype=f32, out=out)
                else:
                    torch.addmm(out, a, b, out_dtype=f32, out=out)
                first = False

    dwh = torch.empty(2, 4 * H, H, device=dev)

    def layer_f32():
        torch.mm(x, w.t(), out=xw)
        torch.mm(d, w, out=dx)
        torch.mm(d.t(), x, out=dw)
        for dirn in (0, 1):
            torch.mm(d[:, dirn * 4 * H:(dirn + 1) * 4 * H].t(), x[:, dirn * H:(dirn + 1) * H],
                     out=dwh[dirn])

    layer_split()
    check = {'xw': errors(xw[sub], x[sub].double() @ w.double().t()),
             'dx': errors(dx[sub], d[sub].double() @ w.double()),
             'dw': errors(dw[sub], d.double().t()[sub] @ x.double()),
             'dwh': errors(dwh[0][sub], d[:, :4 * H].double().t()[sub] @ x[:, :H].double())}
    layer_f32()
    check_f32 = {'xw': errors(xw[sub], x[sub].double() @ w.double().t()),
                 'dx': errors(dx[sub], d[sub].double() @ w.double()),
                 'dw': errors(dw[sub], d.double().t()[sub] @ x.double()),
                 'dwh': errors(dwh[0][sub], d[:, :4 * H].double().t()[sub] @ x[:, :H].double())}
    report['layer'] = {
        'gemms': 'xw, dx, dW_ih, 2 x dW_hh of one C3 layer, 60 layers back to back',
        'f32_ms': round(timed(layer_f32, 60, 10), 3),
        'split_ms': round(timed(layer_split, 60, 10), 3),
        'f32_ms_again': round(timed(layer_f32, 60, 10), 3),
        'split_err': check, 'f32_err': check_f32}
    print(json.dumps(report['layer']), flush=True)
    print(json.dumps(report, indent=1))


if __name__ == '__main__':
    main()
