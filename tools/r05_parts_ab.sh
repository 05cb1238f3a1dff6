#!/bin/bash
# Round 5: CTCASR_WGRAD_PARTS 1 / 2 per workload on one box
mkdir -p gpurun_out; out=gpurun_out/parts_ab.log; : > $out
run() { w=$1; shift; echo "== $w $*" >> $out; env "$@" timeout 400 python bench.py --workload $w --steps 8 --warmup 3 --no-cpu-baseline --no-other-workloads 2>/dev/null > /tmp/b.json; python tools/show_bench.py /tmp/b.json | head -1 | cut -c1-100 >> $out; }
for rep in 1 2; do
for w in "$@"; do
  run $w CTCASR_WGRAD_PARTS=2
  run $w CTCASR_WGRAD_PARTS=1
done
done
cat $out
