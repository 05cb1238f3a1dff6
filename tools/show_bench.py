#!/usr/bin/env python
"""One-line summary of a bench.py JSON line: python tools/show_bench.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
out = ['{}: {} {} ({} ms/step)'.format(d['config'].get('name', '?'), d['value'], d['unit'],
                                      d['ms_per_step'])]
roof = d.get('roofline') or {}
if roof.get('us_per_time_step'):
    out.append('dominant kernel {} us/time step (other pass {})'.format(
        roof['us_per_time_step'], roof.get('other_pass', {}).get('us_per_time_step')))
for name, other in (d.get('other_workloads') or {}).items():
    out.append('{}: {} ({} ms/step)'.format(name, other.get('value', other.get('error')),
                                            other.get('ms_per_step')))
if 'ctc_loss_delta' in d:
    out.append('loss delta {} logits delta {}'.format(d['ctc_loss_delta'],
                                                      d['logits_max_abs_delta']))
print('; '.join(out))
