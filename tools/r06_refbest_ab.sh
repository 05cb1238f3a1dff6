#!/bin/bash
# Round 6: ref_best (3 conv + 4 x BiLSTM-2048, batch 16) with the plain / K-pair backward recurrence,
# alternating on one box; then the full-length tests of the reference's own models
mkdir -p gpurun_out; out=gpurun_out/r06_refbest_ab.log; : > $out
B="python bench.py --workload ref_best --no-cpu-baseline --no-other-workloads --no-parity-probe --steps 6 --warmup 2"
for rep in 1 2; do
  for kp in 0 1; do
    echo "== CTCASR_RNN_KPAIR_2048=$kp" >> $out
    CTCASR_RNN_KPAIR_2048=$kp timeout 600 $B 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['ms_per_step'], 'ms/step;', r.get('kernel','')[:40], r.get('us_per_time_step'), 'us per time step; loss', d.get('loss'))" >> $out
  done
done
cat $out
[ -n "$SKIP_TEST" ] || timeout 2400 python -m pytest tests/test_gpu_model.py -x -q -k "full_length" 2>&1 | tail -5
