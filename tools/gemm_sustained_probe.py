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
    """Power (W) and shader clock (MHz) while a phase runs: every hwmon device of the node is
    sampled from sysfs (a box may expose the other GPUs of its node there as well; the device
    that draws the most is the one under test) and `amd-smi metric` is asked once, mid-phase, for
    the GPU this process sees."""

    def __init__(self, period=0.02):
        self.period, self._stop = period, threading.Event()
        self.devices = []
        for hwmon in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*')):
            power = next((os.path.join(hwmon, n) for n in ('power1_average', 'power1_input')
                          if os.path.exists(os.path.join(hwmon, n))), None)
            clock = os.path.join(hwmon, 'freq1_input')
            if power:
                self.devices.append((hwmon, power, clock if os.path.exists(clock) else None))
        self.samples = {d[0]: [] for d in self.devices}
        self.smi = None

    @staticmethod
    def _amd_smi():
        try:
            out = subprocess.run(['amd-smi', 'metric', '--power', '--clock', '--json'],
                                 capture_output=True, text=True, timeout=10).stdout
            return json.loads(out)
        except Exception as err:
            return {'error': type(err).__name__}

    def __enter__(self):
        self.samples = {d[0]: [] for d in self.devices}
        self.smi, self._stop = None, threading.Event()

        def loop():
            ticks = 0
            while not self._stop.is_set():
                for name, power, clock in self.devices:
                    try:
                        self.samples[name].append((
                            int(open(power).read()) / 1e6,
                            int(open(clock).read()) / 1e6 if clock else None))
                    except (OSError, ValueError):
                        pass
                ticks += 1
                if ticks == 10 and self.smi is None:
                    self.smi = self._amd_smi()
                time.sleep(self.period)
        self.thread = threading.Thread(target=loop, daemon=True)
        self.thread.start()
        return self

    def __exit__(self, *exc):
        self._stop.set()
        self.thread.join()
        return False

    def summary(self):
        best, best_mean = None, -1.0
        for name, rows in self.samples.items():
            power = [p for p, _ in rows]
            if power and sum(power) / len(power) > best_mean:
                best, best_mean = name, sum(power) / len(power)
        out = {'hwmon_devices': len(self.devices), 'amd_smi': self.smi}
        if best is not None:
            rows = self.samples[best]
            clock = [c for _, c in rows if c is not None]
            out.update({'busiest_hwmon': best, 'power_w_mean': round(best_mean, 1),
                        'power_w_max': round(max(p for p, _ in rows), 1), 'samples': len(rows),
                        'sclk_mhz_mean': round(sum(clock) / len(clock), 1) if clock else None})
        return out


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
