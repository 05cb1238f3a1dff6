#!/usr/bin/env python
"""Do tuned library solutions (torch TunableOp over hipBLASLt / rocBLAS) beat the heuristic picks
for the bf16 GEMMs of the split path (split_gemm.py) when they run back to back?  Shapes of one
C3 layer: the forward projection (one call, K = 6 x 2048), one data-gradient piece (K = 8192), one
weight-gradient call of a third of the time steps (K = 6 x 5344 rows), W_hh's likewise.
    python tools/gemm_split_tune_probe.py [csv]"""
import sys
import time

import torch
import torch.cuda.tunable as tunable

F32 = torch.float32


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    csv = sys.argv[1] if len(sys.argv) > 1 else '/tmp/split_tune.csv'
    R, F, G, H = 16000, 2048, 8192, 1024
    bf = dict(device='cuda', dtype=torch.bfloat16)
    xs = torch.randn(R, 6 * F, **bf)
    ws = torch.randn(G, 6 * F, **bf)
    ds = torch.randn(R, 6, G, **bf)
    xw = torch.empty(R, G, device='cuda')
    dx = torch.empty(R, F, device='cuda')
    dw = torch.zeros(G // 2, F, device='cuda')
    dwh = torch.zeros(G // 2, H, device='cuda')
    rows3 = 5344
    cases = {
        'fwd  [16000 x 12288] x [8192 x 12288]^T': (
            lambda: torch.mm(xs, ws.t(), out_dtype=F32, out=xw), 2.0 * R * 6 * F * G),
        'dgrad piece [16000 x 8192] x [8192 x 2048] (+=)': (
            lambda: torch.addmm(dx, ds[:, 0], ws.view(G, 6, F)[:, 0], out_dtype=F32, out=dx),
            2.0 * R * G * F),
        'wgrad W_ih [32064 x 4096]^T x [32064 x 2048] (+=)': (
            lambda: torch.addmm(dw, ds.view(R * 6, G)[:6 * rows3, :G // 2].t(),
                                xs.view(R * 6, F)[:6 * rows3], out_dtype=F32, out=dw),
            2.0 * 6 * rows3 * (G // 2) * F),
        'wgrad W_hh [32064 x 4096]^T x [32064 x 1024] (+=)': (
            lambda: torch.addmm(dwh, ds.view(R * 6, G)[:6 * rows3, :G // 2].t(),
                                xs.view(R * 6, F)[:6 * rows3, :H], out_dtype=F32, out=dwh),
            2.0 * 6 * rows3 * (G // 2) * H)}
    base = {}
    for name, (fn, flops) in cases.items():
        base[name] = timed(fn)
        print('default  {:55s} {:.3f} ms  {:.0f} TF raw'.format(name, base[name],
                                                               flops / base[name] / 1e9), flush=True)
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_max_tuning_duration(150)
    tunable.set_max_tuning_iterations(200)
    tunable.set_filename(csv)
    for name, (fn, flops) in cases.items():
        t0 = time.time()
        fn()
        torch.cuda.synchronize()
        print('tuning {} took {:.1f} s'.format(name, time.time() - t0), flush=True)
    tunable.tuning_enable(False)
    for name, (fn, flops) in cases.items():
        t = timed(fn)
        print('tuned    {:55s} {:.3f} ms  {:.0f} TF raw  ({:+.1f} %)'.format(
            name, t, flops / t / 1e9, 100.0 * (t / base[name] - 1.0)), flush=True)
    for row in tunable.get_results():
        print(row)
    tunable.write_file(csv)


if __name__ == '__main__':
    main()
