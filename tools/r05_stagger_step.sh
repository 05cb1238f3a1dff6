#!/bin/bash
# Round 5: the C3 step with the staggered backward recurrence and the side-stream knobs again
mkdir -p gpurun_out; out=gpurun_out/stagger_step.log; : > $out
run() { echo "== $*" >> $out; env "$@" timeout 300 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads 2>/dev/null > /tmp/b.json; python tools/show_bench.py /tmp/b.json | head -1 | cut -c1-120 >> $out; }
for cfg in "$@"; do run $cfg; done
cat $out
