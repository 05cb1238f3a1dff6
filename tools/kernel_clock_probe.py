#!/usr/bin/env python
"""Effective shader clock of every dispatch of a kernel: GRBM_GUI_ACTIVE / 8 XCDs / duration, both
from ONE `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` database.

    rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/kc -o kc -- python tools/rnn_microbench.py 500 32 1024
    python tools/kernel_clock_probe.py /tmp/kc/.../kc_results.db prnn
"""
import sqlite3
import sys


def main(path, pattern):
    con = sqlite3.connect(path)
    rows = con.execute(
        "select kernel_name, value, start, end from counters_collection where counter_name = "
        "'GRBM_GUI_ACTIVE' and kernel_name like ? order by start", ('%' + pattern + '%',)).fetchall()
    print('| kernel | duration us | GRBM_GUI_ACTIVE | effective clock GHz |')
    print('|---|---:|---:|---:|')
    for name, value, start, end in rows:
        dur = (end - start) / 1e3
        print('| `{}` | {:.1f} | {:.0f} | {:.3f} |'.format(name[:70], dur, value,
                                                        value / 8.0 / (dur * 1e3)))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
