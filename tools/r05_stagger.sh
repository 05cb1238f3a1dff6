#!/bin/bash
# Round 5: the staggered-tile backward recurrence (prnn_bwd16s_kernel) - parity test, then the
# kernel alone against the one-barrier kernel (tools/rnn_microbench.py), then the C3 step.
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "staggered" > gpurun_out/stagger_test.log 2>&1
echo "rc=$?" >> gpurun_out/stagger_test.log
tail -5 gpurun_out/stagger_test.log
for st in "" 1; do
  echo "== CTCASR_STAGGER=$st" >> gpurun_out/stagger_micro.log
  CTCASR_F16=1 CTCASR_XCD=1 CTCASR_STAGGER=$st timeout 300 python tools/rnn_microbench.py 500 32 1024 >> gpurun_out/stagger_micro.log 2>&1
done
cat gpurun_out/stagger_micro.log
