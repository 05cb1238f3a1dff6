#!/usr/bin/env python
"""Evidence for the "sustained fp32-MFMA rate" DESIGN.md section 5 argues from: the C3 input
projection (16000 x 2048 x 8192, fp32, hipBLASLt through torch.mm)

  burst      after 0.5 s of idle: the first launches, one HIP event pair per launch
  sustained  200 launches back to back, one event pair per launch (means over windows)
  after_fwd  right behind a whole-chip persistent forward recurrence (256 CUs, 3 ms)
  after_bwd  right behind a half-chip persistent backward recurrence (128 CUs, 6 ms)
  beside_bwd on the side stream WHILE the half-chip backward recurrence runs (the C3 step's
             weight-gradient situation)

with the GPU's power and shader clock sampled from the driver while each phase runs (sysfs hwmon
when readable, else `amd-smi metric`).  Under `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace`
the same script gives the cycle count of every GEMM dispatch; `--analyze <db>` turns that into
an effective clock per dispatch (GRBM_GUI_ACTIVE / 8 XCDs / duration).

    python tools/gemm_sustained_probe.py                       # timings + power / clock samples
    rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/gp -o gp -- \
        python tools/gemm_sustained_probe.py --phases burst,sustained --reps 60
    python tools/gemm_sustained_probe.py --analyze /tmp/gp/.../gp_results.db
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ROWS, K, N = 16000, 2048, 8192
FLOPS = 2.0 * ROWS * K * N


class Sampler:
    """Power (W) and shader clock (MHz) of GPU 0, sampled every ``period`` seconds."""

    def __init__(self, period=0.02):
        self.period, self.samples, self._stop = period, [], threading.Event()
        self.power_file = self.clock_file = None
        for hwmon in glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'):
            for name in ('power1_average', 'power1_input'):
                if os.path.exists(os.path.join(hwmon, name)) and self.power_file is None:
                    self.power_file = os.path.join(hwmon, name)
            if os.path.exists(os.path.join(hwmon, 'freq1_input')) and self.clock_file is None:
                self.clock_file = os.path.join(hwmon, 'freq1_input')
        self.source = 'sysfs hwmon' if self.power_file else 'amd-smi metric'
        if not self.power_file:
            self.period = max(period, 0.25)

    def _read(self):
        if self.power_file:
            try:
                power = int(open(self.power_file).read()) / 1e6
                clock = int(open(self.clock_file).read()) / 1e6 if self.clock_file else None
                return power, clock
            except (OSError, ValueError):
                return None, None
        try:
            out = subprocess.run(['amd-smi', 'metric', '-g', '0', '--power', '--clock', '--json'],
                                 capture_output=True, text=True, timeout=5).stdout
            data = json.loads(out)
            data = data[0] if isinstance(data, list) else data
            if 'gpu_data' in data:
                data = data['gpu_data'][0]
            power = data.get('power', {}).get('socket_power', {})
            power = power.get('value') if isinstance(power, dict) else power
            clock = data.get('clock', {}).get('gfx_0', {}).get('clk', {})
            clock = clock.get('value') if isinstance(clock, dict) else clock
            return (float(power) if power not in (None, 'N/A') else None,
                    float(clock) if clock not in (None, 'N/A') else None)
        except Exception:
            return None, None

    def __enter__(self):
        self.samples, self._stop = [], threading.Event()

        def loop():
            while not self._stop.is_set():
                self.samples.append(self._read())
                time.sleep(self.period)
        self.thread = threading.Thread(target=loop, daemon=True)
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self.thread.join()
        return False

    def summary(self):
        power = [p for p, _ in self.samples if p is not None]
        clock = [c for _, c in self.samples if c is not None]
        fmt = lambda xs: None if not xs else {'mean': round(sum(xs) / len(xs), 1),
                                              'max': round(max(xs), 1), 'n': len(xs)}
        return {'power_w': fmt(power), 'sclk_mhz': fmt(clock), 'source': self.source}


def analyze(path):
    import sqlite3
    con = sqlite3.connect(path)
    cols = [r[1] for r in con.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    ccols = [r[1] for r in con.execute('pragma table_info(counters_collection)')]
    print('counters_collection columns:', ccols)
    rows = con.execute("select {n}, start, end from kernels where {n} like 'Cijk%' order by start"
                       .format(n=name_col)).fetchall()
    order = 'start' if 'start' in ccols else ('dispatch_id' if 'dispatch_id' in ccols else 'rowid')
    vals = con.execute("select value from counters_collection where counter_name = "
                       "'GRBM_GUI_ACTIVE' and kernel_name like 'Cijk%' order by {}".format(order)
                       ).fetchall()
    print('{} GEMM dispatches, {} counter rows'.format(len(rows), len(vals)))
    print('| dispatch | gap before us | duration us | GRBM_GUI_ACTIVE | effective clock GHz | TFLOP/s |')
    print('|---:|---:|---:|---:|---:|---:|')
    last_end = None
    for i, ((name, start, end), (value,)) in enumerate(zip(rows, vals)):
        dur = (end - start) / 1e3
        gap = (start - last_end) / 1e3 if last_end else 0.0
        last_end = end
        if i < 12 or i % 10 == 0 or gap > 1e4:
            print('| {} | {:.0f} | {:.1f} | {:.0f} | {:.3f} | {:.1f} |'.format(
                i, gap, dur, value, value / 8.0 / (dur * 1e3), FLOPS / (dur * 1e-6) / 1e12))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--phases', default='burst,sustained,after_fwd,after_bwd,beside_bwd')
    ap.add_argument('--reps', type=int, default=200)
    ap.add_argument('--analyze')
    args = ap.parse_args()
    if args.analyze:
        return analyze(args.analyze)
    import numpy as np
    import torch
    from ctc_asr_amd import hip
    hip.load()
    x = torch.randn(ROWS, K, device='cuda')
    w = torch.randn(N, K, device='cuda') / K ** 0.5
    out = torch.empty(ROWS, N, device='cuda')
    gemm = lambda: torch.mm(x, w.t(), out=out)

    def timed(count, before=None):
        pairs = []
        if before is not None:
            before()
        for _ in range(count):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); gemm(); b.record()
            pairs.append((a, b))
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in pairs]

    tf = lambda ms: round(FLOPS / (ms * 1e-3) / 1e12, 1)
    sampler = Sampler()
    for _ in range(3):
        gemm()
    torch.cuda.synchronize()
    report = {'gemm': '{} x {} x {} fp32 (C3 input projection)'.format(ROWS, K, N)}
    phases = args.phases.split(',')
    if 'burst' in phases:
        firsts = []
        for _ in range(6):
            time.sleep(0.5)
            firsts.append(timed(4))
        med = np.median(np.array(firsts), axis=0)
        report['burst'] = {'ms_launch_1_to_4_median_of_6': [round(float(v), 3) for v in med],
                           'tflops': [tf(v) for v in med]}
    if 'sustained' in phases:
        time.sleep(0.5)
        with sampler:
            ms = timed(args.reps)
        win = lambda a, b: round(float(np.mean(ms[a:b])), 3)
        n = len(ms)
        report['sustained'] = {
            'launches': n, 'ms_first_3': [round(v, 3) for v in ms[:3]],
            'ms_mean_10_20': win(10, 20), 'ms_mean_mid': win(n // 2 - 10, n // 2 + 10),
            'ms_mean_last_50': win(n - 50, n), 'tflops_last_50': tf(win(n - 50, n)),
            'tflops_first': tf(ms[0]), **sampler.summary()}
    if any(p in phases for p in ('after_fwd', 'after_bwd', 'beside_bwd')):
        T, B, H = 500, 32, 1024
        g = torch.Generator(device='cuda').manual_seed(0)
        xw = torch.randn(T, B, 2, 4 * H, device='cuda', generator=g) * 0.5
        w_hh = torch.randn(2, 4 * H, H, device='cuda', generator=g) / 32
        dy = torch.randn(T, B, 2 * H, device='cuda', generator=g)
        wt = hip.transpose_batched(w_hh)
        y, reserve, ws = hip.rnn_fwd('lstm', xw, w_hh)
        dxw = hip.rnn_bwd('lstm', dy, y, wt, reserve, workspace=ws)
        torch.cuda.synchronize()
        fwd = lambda: hip.rnn_fwd('lstm', xw, w_hh, y=y, reserve=reserve, workspace=ws)
        bwd = lambda: hip.rnn_bwd('lstm', dy, y, wt, reserve, dxw=dxw, workspace=ws)
        for name, before in (('after_fwd', fwd), ('after_bwd', bwd)):
            if name not in phases:
                continue
            runs = []
            for _ in range(8):          # steady alternation, like layers of a training step
                runs.append(timed(3, before))
            med = np.median(np.array(runs[2:]), axis=0)
            report[name] = {'ms_launch_1_to_3_median': [round(float(v), 3) for v in med],
                            'tflops': [tf(v) for v in med]}
        if 'beside_bwd' in phases:
            side = torch.cuda.Stream()
            runs = []
            for _ in range(8):
                torch.cuda.synchronize()
                ticket = 1 + len(runs)
                ready = torch.cuda.Event()
                ready.record()
                hip.rnn_bwd('lstm', dy, y, wt, reserve, dxw=dxw, workspace=ws, ticket=ticket)
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    hip.rnn_resident_gate('lstm', ws, T, B, H, ticket, 300)
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    # ONE GEMM that ends before the recurrence (5.6 ms) does: it only ever sees
                    # the 128 CUs the recurrence leaves free
                    a.record(side)
                    torch.mm(x[:ROWS // 2], w.t(), out=out[:ROWS // 2])     # half the rows
                    b.record(side)
                torch.cuda.synchronize()
                runs.append(a.elapsed_time(b))
            med = float(np.median(runs[2:]))
            report['beside_bwd'] = {'gemm': '{} x {} x {} (half the rows)'.format(ROWS // 2, K, N),
                                    'ms_per_gemm_median': round(med, 3),
                                    'tflops_on_128_cus': round(tf(med) / 2, 1),
                                    'frac_of_half_chip_peak': round(tf(med) / 2 / (157.3 / 2), 3)}
        hip.rnn_poll_error('lstm', ws, T, B, H)
    print(json.dumps(report, indent=1))


if __name__ == '__main__':
    main()
