#!/bin/bash
# the reference's own default model (3 convs + 4 x BiRNN-ReLU-2048, batch 16) and its best published
# one (LSTM-2048): where the step goes with the round-4 kernels
mkdir -p gpurun_out
for w in ref_default ref_best; do
  python bench.py --workload $w --steps 8 --warmup 3 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r04_$w.json 2> gpurun_out/r04_$w.err
  python tools/show_bench.py gpurun_out/r04_$w.json
  bash tools/kt_only.sh $w kt_$w > gpurun_out/kt_$w.log 2>&1
  head -34 gpurun_out/kt_$w/kt.md | cut -c1-160
done
