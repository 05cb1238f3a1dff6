#!/usr/bin/env python
"""Assemble a committed profile summary from what tools/prof_round.sh leaves under gpurun_out/:

    python tools/make_profile_md.py gpurun_out/r02/prof_c3_a c3 profiles/r02_c3_kernel_trace_and_pmc.md

and record the PMC traffic of the dominant recurrence kernels in profiles/pmc_traffic.json (the
file bench.py reads ``roofline.traffic`` from), keyed by workload, pass and steps per launch.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TRAFFIC_JSON = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')


def pmc_rows(table):
    """[(kernel, counter, dispatches, avg)] of a prof_summary.py --pmc table."""
    rows = []
    for line in table.splitlines():
        cells = [c.strip() for c in line.strip().strip('|').split('|')]
        if len(cells) == 4 and cells[2].isdigit():
            rows.append((cells[0].strip('`'), cells[1], int(cells[2]), float(cells[3])))
    return rows


def value(rows, kernel, counter):
    """Average of ``counter`` over the dispatches of the kernel named EXACTLY ``kernel``."""
    for name, ctr, _, avg in rows:
        if name == kernel and ctr == counter:
            return avg
    return None


def step_kernel(rows, pattern):
    """Full name of the kernel matching ``pattern`` that the timed steps ran: the one with the
    most dispatches (the parity probe's other arithmetic variants launch a handful each; VERDICT
    r04: a substring match picked the fp32 kernel's counters for the fp16 kernel's row)."""
    best = None
    for name, _, dispatches, _ in rows:
        if pattern in name and (best is None or dispatches > best[1]):
            best = (name, dispatches)
    return best[0] if best else None


def algorithmic_bytes(which, batch, hidden, gates, steps):
    """HBM bytes one launch of the persistent recurrence must move (both directions), fp32."""
    row = batch * hidden * 4
    if which == 'rnn_bwd':       # dy, gates, c_t + c_t-1 in; dxw + exchange (dgates) out
        per = row * (1 + gates + (2 if gates == 4 else 1)) + 2 * gates * row
    else:                        # xw in; y, reserve (gates + c), exchange (h) out
        per = gates * row + row * (1 + (gates + 1 if gates == 4 else 0) + 1)
    return 2 * steps * per + 2 * gates * hidden * hidden * 4


def main(src, workload, out_md):
    import bench
    read = lambda name: open(os.path.join(src, name)).read()
    line = json.loads(read('bench.json').strip().splitlines()[-1])
    kt_line = json.loads(read('kt_bench.json').strip().splitlines()[-1])
    fetch, write, sq = pmc_rows(read('fetch.md')), pmc_rows(read('write.md')), pmc_rows(read('sq.md'))
    filters, layers, hidden, dense, batch, seconds, cell = bench.WORKLOADS[workload]
    gates = {'lstm': 4, 'gru': 3}.get(cell, 1)
    # (c5: a bucket sequence - mean steps per launch of the backward pass, two launches per layer)
    t_out = line['config'].get('ctc_steps') or \
        round(2 * line['roofline']['mean_time_steps_per_launch'])
    timeline = [l for l in read('timeline.txt').splitlines()
                if not any(k in l for k in ('elementwise_kernel', 'copyBuffer', 'SubTensor',
                                            'batched_transpose', 'vectorized_elementwise',
                                            'reduce_kernel', 'transpose_kernel'))]
    entries, traffic_text = [], []
    for which, pattern in (('rnn_bwd', 'prnn_bwd'), ('rnn_fwd', 'prnn_fwd')):
        kernel = step_kernel(fetch, pattern)
        if kernel is None and cell in ('rnn_relu', 'rnn_tanh'):
            # the ReLU cell's fp16-pipe kernel: one template, <true> = backward
            kernel = step_kernel(fetch, 'prnn_relu16_kernel<{}>'.format(
                'true' if which == 'rnn_bwd' else 'false'))
        if kernel is None:
            continue
        f_kb, w_kb = value(fetch, kernel, 'FETCH_SIZE'), value(write, kernel, 'WRITE_SIZE')
        if f_kb is None or w_kb is None:
            continue
        # launches per layer-pass as the model cuts them (bench's own roofline for the dominant
        # pass, the model defaults for the other)
        dominant = line['roofline']['kernel'].startswith('prnn_' + which[4:])
        if 'mean_time_steps_per_launch' in line['roofline']:
            steps = line['roofline']['mean_time_steps_per_launch'] * (1 if dominant else 2)
        else:
            steps = round(line['roofline']['algorithmic_flops_per_launch'] /
                          (2.0 * 2 * batch * hidden * gates * hidden)) if dominant else t_out
        alg = algorithmic_bytes(which, batch, hidden, gates, steps)
        busy = value(sq, kernel, 'SQ_VALU_MFMA_BUSY_CYCLES')
        gui = value(sq, kernel, 'GRBM_GUI_ACTIVE')
        entries.append({'workload': workload, 'pass': which, 'steps_per_launch': steps,
                        'kernel': kernel, 'fetch_kb': f_kb, 'write_kb': w_kb,
                        'algorithmic_bytes': alg, 'source': os.path.relpath(out_md, ROOT)})
        raw, corrected = (f_kb + w_kb) * 1024, (2 * f_kb + w_kb) * 1024
        text = ('`{}` (one launch = {} time steps x 2 directions): FETCH_SIZE {:.0f} KB + '
                'WRITE_SIZE {:.0f} KB = {:.0f} MB as reported; with the gfx950 FETCH_SIZE x2 '
                'correction (MI355X_MICROARCH.md, section HBM) {:.0f} MB.  Algorithmic HBM bytes '
                'of such a launch: {:.0f} MB -> traffic / algorithmic = {:.2f} (corrected), '
                '{:.2f} (raw).'.format(kernel[:60], steps, f_kb, w_kb, raw / 1e6, corrected / 1e6,
                                       alg / 1e6, corrected / alg, raw / alg))
        if busy and gui:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8 = cycles the kernel ran (at the clock
            # it really had); busy cycles are summed over the SIMDs of the CUs it occupies
            cus = 256 if which == 'rnn_fwd' else 128
            if 'rnn_bwd_whole_chip' in line.get('config', {}).get('workload', '') or \
                    (hidden == 2048 and cell in ('lstm', 'gru')):
                cus = 256       # (H = 2048 LSTM / GRU: one direction per launch on the whole chip)
            text += ('  MFMA-busy: SQ_VALU_MFMA_BUSY_CYCLES {:.0f} M / (GRBM_GUI_ACTIVE {:.1f} M '
                     '/ 8 XCDs x {} CUs x 4 SIMDs) = {:.0f} % of the CUs the kernel occupies.'
                     .format(busy / 1e6, gui / 1e6, cus, 100.0 * busy / (gui / 8 * cus * 4)))
        traffic_text.append(text)

    out = ['# {}: bench workload `{}` - rocprofv3 kernel trace + PMC\n'.format(
               os.path.basename(out_md)[:-3], workload),
           line['config']['workload'] + '.  All on one MI355X through gpurun; counters in their '
           'own passes with `--kernel-trace` only (`tools/prof_round.sh {} <out>`):\n```\n'
           'rocprofv3 --kernel-trace --stats ... -- python bench.py --workload {w} --steps 4 '
           '--warmup 2 --no-cpu-baseline --no-other-workloads\n'
           'rocprofv3 --pmc FETCH_SIZE --kernel-trace ...   (--steps 2 --warmup 2)\n'
           'rocprofv3 --pmc WRITE_SIZE --kernel-trace ...\n'
           'rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA '
           'SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE '
           '--kernel-trace ...\n```\n'.format(workload, w=workload),
           'Unprofiled bench line of the same build (`--steps 8 --warmup 3`):\n```\n' +
           json.dumps(line) + '\n```\n',
           'Dominant kernel, HIP event pairs recorded by the library around each launch '
           '(`rnn_kernel_events`): **{} us** per launch unprofiled, {} us in the profiled run '
           '(kernel trace row below).\n'.format(line['roofline']['avg_launch_us'],
                                                kt_line['roofline']['avg_launch_us']),
           '## Kernel trace, steady state\n\n' + read('kt.md') +
           '\n(`resident_gate_kernel`: a one-lane gate on the side stream, spin bounded at 300 us.  '
           'Its interval in the trace starts when the queue reaches its packet, so a gate that '
           'sits behind a `hipStreamWaitEvent` on the main stream shows the wait for that event - '
           'the 3.4 ms entries coincide with the 3.9 ms data-gradient GEMM of the main stream in '
           'the timeline below - the others take 5 - 9 us.)\n',
           '\n## PMC FETCH_SIZE (KB per dispatch, as reported)\n\n' + read('fetch.md'),
           '\n## PMC WRITE_SIZE (KB per dispatch, as reported)\n\n' + read('write.md'),
           '\n## Traffic and MFMA-busy of the recurrence kernels\n\n' + '\n\n'.join(traffic_text) +
           '\n\nThe excess over the algorithmic bytes is the exchange buffer: every workgroup of a '
           'direction re-reads the whole published block each step; what misses the XCD L2s shows '
           'up as FETCH_SIZE (Infinity-Cache hits included).  The kernel is bound by the per-CU '
           'load path and the exchange latency, not by HBM bandwidth (DESIGN.md section 4.1).\n',
           '\n## PMC SQ counters\n\n' + read('sq.md'),
           '\n## Timeline of the last training step (queue 1 = main stream, queue 2 = '
           'weight-gradient side stream; small elementwise kernels omitted)\n\n```\n' +
           '\n'.join(timeline) + '\n```\n']
    with open(out_md, 'w') as handle:
        handle.write('\n'.join(out))
    table = {'entries': []}
    if os.path.exists(TRAFFIC_JSON):
        table = json.load(open(TRAFFIC_JSON))
    keep = [e for e in table['entries']
            if not any(e['workload'] == n['workload'] and e['pass'] == n['pass'] for n in entries)]
    table['entries'] = keep + entries
    table['note'] = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, KB per dispatch as reported; '
                     'written by tools/make_profile_md.py, read by bench.py (roofline.traffic)')
    with open(TRAFFIC_JSON, 'w') as handle:
        json.dump(table, handle, indent=1)
    print(out_md, [(e['pass'], e['steps_per_launch'], e['fetch_kb'], e['write_kb']) for e in entries])


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2], sys.argv[3])
