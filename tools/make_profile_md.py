#!/usr/bin/env python
"""Assemble profiles/r01_final_c2_kernel_trace_and_pmc.md from the small summaries that
tools/prof_final.sh leaves under gpurun_out/final/ (bench line, kernel trace, PMC passes,
timeline)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(ROOT, 'gpurun_out', 'final')


def read(name):
    with open(os.path.join(O, name)) as handle:
        return handle.read()


def pmc_value(table, kernel):
    for line in table.splitlines():
        if kernel in line:
            return float(line.split('|')[-2])
    return None


def main():
    bench = read('bench.json').strip().splitlines()[-1]
    d = json.loads(bench)
    kt_bench = json.loads(read('kt_bench.json').strip().splitlines()[-1])
    fetch, write = read('fetch.md'), read('write.md')
    timeline = [l for l in read('timeline.txt').splitlines()
                if not any(k in l for k in ('elementwise_kernel', 'fillBuffer', 'copyBuffer',
                                            'SubTensor', 'batched_transpose',
                                            'vectorized_elementwise', 'reduce_kernel'))]
    f_kb, w_kb = pmc_value(fetch, 'prnn_bwd_kernel'), pmc_value(write, 'prnn_bwd_kernel')
    out = ['# r01 final: C2 bench (DS2 2-conv + 2xBiLSTM-1024, B=16, 10 s) - rocprofv3 kernel '
           'trace + PMC\n',
           'All on one MI355X through gpurun; counters in their own passes with `--kernel-trace` '
           'only (`tools/prof_final.sh`):\n```\n'
           'rocprofv3 --kernel-trace --stats -d /tmp/pf/kt -o c2 -- python bench.py --steps 10 '
           '--warmup 3 --no-cpu-baseline\n'
           'rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf/fetch -o c2 -- python bench.py '
           '--steps 2 --warmup 3 --no-cpu-baseline\n'
           'rocprofv3 --pmc WRITE_SIZE --kernel-trace ...   (same)\n'
           'rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA '
           'SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE '
           '--kernel-trace ...\n'
           'python tools/prof_summary.py <db> 45 5        # steady state = last 5 training steps\n'
           'python tools/prof_summary.py --pmc <db>\n'
           'python tools/prof_summary.py --timeline <db>  # launch order per queue, last step\n'
           '```\n',
           'Unprofiled bench line of the same build (`python bench.py --steps 10 --warmup 3`, '
           'incl. cpu_baseline):\n```\n' + bench + '\n```\n',
           'Dominant kernel, HIP event pairs recorded by the library right around each launch '
           '(`rnn_kernel_events`): **{} us** per launch unprofiled, {} us in the profiled run '
           '(its kernel trace: `prnn_bwd_kernel` row below).\n'.format(
               d['roofline']['avg_launch_us'], kt_bench['roofline']['avg_launch_us']),
           '## Kernel trace, steady state\n\n' + read('kt.md'),
           '\n## PMC FETCH_SIZE (KB per dispatch, as reported)\n\n' + fetch,
           '\n## PMC WRITE_SIZE (KB per dispatch, as reported)\n\n' + write]
    if f_kb and w_kb:
        out.append("""
`roofline.traffic` for `prnn_bwd_kernel` (one launch = 167 of 500 time steps) = ({:.0f} + {:.0f}) KB
x 1024 = {:.0f} MB.  Algorithmic HBM bytes of such a launch, per step and direction (B = 16,
H = 1024, fp32): read dy 64 KB + gates 256 KB + cells (c_t, c_t-1) 128 KB, write dxw 256 KB +
exchange (dgates) 256 KB = 960 KB; x 2 directions x 167 steps = 321 MB, + 2 x 16 MB recurrent
weights loaded once = 353 MB.  The difference is the exchange buffer, which all 64 workgroups of a
direction re-read every step (16 MB per step and direction at the L2s; what misses L2 shows up as
FETCH_SIZE).  The kernel is latency-, not bandwidth-bound (DESIGN.md section 4.1).
""".format(f_kb, w_kb, (f_kb + w_kb) * 1024 / 1e6))
    out.append('\n## PMC SQ counters\n\n' + read('sq.md'))
    out.append('\n## Timeline of the last training step (queue 1 = main stream, queue 2 = '
               'weight-gradient side stream; small elementwise kernels omitted)\n\n```\n' +
               '\n'.join(timeline) + '\n```\n')
    path = os.path.join(ROOT, 'profiles', 'r01_final_c2_kernel_trace_and_pmc.md')
    with open(path, 'w') as handle:
        handle.write('\n'.join(out))
    print(path, f_kb, w_kb)


if __name__ == '__main__':
    main()
