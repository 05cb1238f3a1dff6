#!/bin/bash
# Round 6: the price of a pair hand-off (tools/pair_handoff) + the staggered kernel on this box
mkdir -p gpurun_out; out=gpurun_out/r06_handoff.log; : > $out
timeout 120 tools/pair_handoff/probe >> $out 2>&1
echo "rc $?" >> $out
for st in 1; do
  CTCASR_F16=1 CTCASR_XCD=1 CTCASR_STAGGER=$st timeout 300 python tools/rnn_microbench.py 500 32 1024 2>&1 | grep "bwd\|phases\|checksum" >> $out
  CTCASR_F16=1 CTCASR_XCD=1 CTCASR_STAGGER=$st CTCASR_RNN_PROF=1 timeout 300 python tools/rnn_microbench.py 500 32 1024 2>&1 | grep "bwd\|phases\|checksum" >> $out
done
cat $out
