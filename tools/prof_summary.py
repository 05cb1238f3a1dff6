#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (``*_results.db``) into a per-kernel table:
calls, total / average / min / max duration, share of GPU kernel time.  Used to produce the
``profiles/*.md`` summaries committed with each round.

    python tools/prof_summary.py gpurun_out/prof_x/c2_results.db > profiles/r01_x.md
"""

import sqlite3
import sys


def main(path, top=40, last_steps=0):
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    where = ''
    if last_steps:
        # a training step ends with the Adam launch: keep the last `last_steps` whole steps
        ends = [r[0] for r in cur.execute(
            "select end from kernels where {n} like '%adam_kernel%' order by end".format(
                n=name_col))]
        if len(ends) > last_steps:
            where = 'where start > {} and end <= {}'.format(ends[-last_steps - 1], ends[-1])
            print('window: last {} training steps, {:.3f} ms wall\n'.format(
                last_steps, (ends[-1] - ends[-last_steps - 1]) / 1e6))
    rows = cur.execute(
        'select {n}, count(*), sum(end - start), avg(end - start), min(end - start), '
        'max(end - start) from kernels {w} group by {n} order by 3 desc'.format(
            n=name_col, w=where)).fetchall()
    total = sum(r[2] for r in rows) or 1
    print('| kernel | calls | total ms | avg us | min us | max us | % |')
    print('|---|---:|---:|---:|---:|---:|---:|')
    for name, calls, tot, avg, mn, mx in rows[:top]:
        short = name if len(name) <= 110 else name[:107] + '...'
        print('| `{}` | {} | {:.3f} | {:.2f} | {:.2f} | {:.2f} | {:.1f} |'.format(
            short, calls, tot / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    print('\ntotal kernel time: {:.3f} ms over {} dispatches'.format(
        total / 1e6, sum(r[1] for r in rows)))




def timeline(path):
    """Launch-ordered timeline of the last training step: start offset, duration, queue, gap to
    the previous kernel of the same queue.  Shows where a stream idles."""
    con = sqlite3.connect(path)
    cur = con.cursor()
    cols = [r[1] for r in cur.execute('pragma table_info(kernels)')]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    queue_col = 'queue_id' if 'queue_id' in cols else ('stream_id' if 'stream_id' in cols
                                                      else 'tid')
    ends = [r[0] for r in cur.execute(
        "select end from kernels where {n} like '%adam_kernel%' order by end".format(n=name_col))]
    t0, t1 = ends[-2], ends[-1]
    rows = cur.execute('select start, end, {q}, {n} from kernels where start > ? and end <= ? '
                       'order by start'.format(q=queue_col, n=name_col), (t0, t1)).fetchall()
    last_end = {}
    busy = {}
    print('step wall {:.3f} ms; columns: start us | dur us | queue | gap us | kernel'.format(
        (t1 - t0) / 1e6))
    for start, end, queue, name in rows:
        gap = (start - last_end.get(queue, t0)) / 1e3
        last_end[queue] = end
        busy[queue] = busy.get(queue, 0) + end - start
        print('{:9.1f} {:8.1f} {:>4} {:7.1f}  {}'.format((start - t0) / 1e3, (end - start) / 1e3,
                                                       queue, gap, name[:70]))
    for queue, total in busy.items():
        print('queue {}: busy {:.3f} ms'.format(queue, total / 1e6))


def pmc(path, top=12):
    """Per-kernel average of every collected PMC counter (rocprofv3 --pmc ... --kernel-trace)."""
    con = sqlite3.connect(path)
    rows = con.execute(
        'select kernel_name, counter_name, count(*), avg(value) from counters_collection '
        'group by kernel_name, counter_name order by sum(value) desc').fetchall()
    print('| kernel | counter | dispatches | avg per dispatch |')
    print('|---|---|---:|---:|')
    for name, counter, calls, avg in rows[:top]:
        short = name if len(name) <= 100 else name[:97] + '...'
        print('| `{}` | {} | {} | {:.1f} |'.format(short, counter, calls, avg))


if __name__ == '__main__':
    if sys.argv[1] == '--timeline':
        timeline(sys.argv[2])
    elif sys.argv[1] == '--pmc':
        pmc(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 12)
    else:
        main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40,
             int(sys.argv[3]) if len(sys.argv) > 3 else 0)
