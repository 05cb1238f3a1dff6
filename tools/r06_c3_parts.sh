#!/bin/bash
# Round 6: C3 with the weight-gradient tiles in 1 / 2 / 3 parts, alternating on one box
mkdir -p gpurun_out; out=gpurun_out/r06_c3_parts.log; : > $out
B="python bench.py --workload c3 --no-cpu-baseline --no-other-workloads --no-parity-probe --steps 12 --warmup 3"
for rep in 1 2; do
  for parts in 2 1 3; do
    echo "== CTCASR_WGRAD_PARTS=$parts" >> $out
    CTCASR_WGRAD_PARTS=$parts timeout 600 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['ms_per_step'], 'ms/step; bwd recurrence', r.get('us_per_time_step'), 'us per time step')" >> $out
  done
done
cat $out
