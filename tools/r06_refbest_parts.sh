#!/bin/bash
# Round 6: ref_best, the weight-gradient tiles in 1 / 2 / 3 parts (they run alone on the whole chip there)
mkdir -p gpurun_out; out=gpurun_out/r06_refbest_parts.log; : > $out
B="python bench.py --workload ref_best --no-cpu-baseline --no-other-workloads --no-parity-probe --steps 6 --warmup 2"
for rep in 1 2; do
  for parts in 2 1 3; do
    echo "== CTCASR_WGRAD_PARTS=$parts" >> $out
    CTCASR_WGRAD_PARTS=$parts timeout 600 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], 'ms/step; loss', d.get('loss'))" >> $out
  done
done
cat $out
