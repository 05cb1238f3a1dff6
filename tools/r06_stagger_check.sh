#!/bin/bash
# Round 6: the staggered kernels after the cache-line fix: tests, then the kernel alone (B = 32)
mkdir -p gpurun_out; out=gpurun_out/r06_stagger_check.log; : > $out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "staggered or k_pairs" 2>&1 | tail -6 >> $out
for st in 1 1; do CTCASR_F16=1 CTCASR_XCD=1 CTCASR_STAGGER=$st timeout 300 python tools/rnn_microbench.py 500 32 1024 2>&1 | grep "bwd\|checksum" >> $out; done
cat $out
