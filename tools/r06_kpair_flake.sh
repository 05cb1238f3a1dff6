#!/bin/bash
# Round 6: how often, and where, does test_rnn_bwd_k_pairs fail?
mkdir -p gpurun_out; out=gpurun_out/r06_kpair_flake.log; : > $out
for i in $(seq 1 ${1:-12}); do
  timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -k "k_pairs and not 2048" 2>&1 | grep "^tests/test_gpu_kernels.py:[0-9]*: \|^FAILED\|passed" | tr '\n' ' ' >> $out; echo >> $out
done
cat $out
