#!/bin/bash
# Round 5: the new defaults (staggered tiles, 2-part weight-gradient tiles, early data-gradient half)
# against the old ones on ONE box, per workload
mkdir -p gpurun_out; out=gpurun_out/defaults_ab.log; : > $out
run() { w=$1; shift; echo "== $w $*" >> $out; env "$@" timeout 300 python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads 2>/dev/null > /tmp/b.json; python tools/show_bench.py /tmp/b.json | head -1 | cut -c1-100 >> $out; }
OLD="CTCASR_RNN_STAGGER=0 CTCASR_WGRAD_PARTS=1 CTCASR_DGRAD_EARLY=0"
for rep in 1 2; do
for w in c3 c2; do
  run $w A=1
  run $w $OLD
  run $w CTCASR_DGRAD_EARLY=0
  run $w CTCASR_WGRAD_PARTS=1
done
done
cat $out
