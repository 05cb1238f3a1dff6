#!/usr/bin/env python
"""Register / LDS / spill table of every kernel in one .hip file (hipcc
-Rpass-analysis=kernel-resource-usage):  python tools/kernel_resources.py rnn_persistent.hip"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(name):
    src = os.path.join(ROOT, 'ctc_asr_amd', 'csrc', name)
    out = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17',
                          '-fPIC', '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o',
                          '/dev/null'], capture_output=True, text=True).stderr
    cur, rows = None, {}
    for line in out.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = m.group(1)
            rows[cur] = {}
            continue
        m = re.search(r'remark:\s+([^:]+): (\S+)', line)
        if m and cur:
            rows[cur][m.group(1).strip()] = m.group(2)
    demangle = subprocess.run(['c++filt'] + list(rows),
                              capture_output=True, text=True).stdout.splitlines()
    print('{:<70s} {:>5s} {:>5s} {:>5s} {:>4s} {:>8s} {:>7s} {:>7s}'.format(
        'kernel', 'VGPR', 'AGPR', 'SGPR', 'occ', 'scratch', 'sspill', 'vspill'))
    for (key, v), nice in zip(rows.items(), demangle):
        nice = re.sub(r'\(anonymous namespace\)::|void |\(.*$', '', nice)
        print('{:<70s} {:>5s} {:>5s} {:>5s} {:>4s} {:>8s} {:>7s} {:>7s}'.format(
            nice[:70], v.get('VGPRs', '?'), v.get('AGPRs', '?'), v.get('TotalSGPRs', '?'),
            v.get('Occupancy [waves/SIMD]', '?'), v.get('ScratchSize [bytes/lane]', '?'),
            v.get('SGPRs Spill', '?'), v.get('VGPRs Spill', '?')))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'rnn_persistent.hip')
