#!/bin/bash
# Round 6: soak runs of the round's build (deferred checks on): no time-out word, no dropped step, no hang
mkdir -p gpurun_out
for spec in "c3 400" "ref_best 150" "c2 600" "ref_default 200"; do set -- $spec
  timeout 900 python bench.py --workload $1 --steps $2 --warmup 5 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r06_soak_$1.json 2>gpurun_out/r06_soak_$1.err; echo "$1 rc=$?"; python tools/show_bench.py gpurun_out/r06_soak_$1.json | cut -c1-160
done
