#!/bin/bash
# Round 5: tuning builds of prnn_bwd16s_kernel on the recurrence microbench (B = 32, T = 500)
mkdir -p gpurun_out; out=gpurun_out/stagger_ab.log; : > $out
run() { echo "== $*" >> $out; env "$@" CTCASR_F16=1 CTCASR_XCD=1 timeout 300 python tools/rnn_microbench.py 500 32 1024 2>&1 | grep "bwd\|phases\|checksum" >> $out; }
run CTCASR_STAGGER=
run CTCASR_STAGGER=1
run CTCASR_STAGGER=1 CTCASR_RNN_PROF=1
for v in "$@"; do
  lib=ctc_asr_amd/csrc/_obj/alt_rnn_persistent_$v.so
  run CTCASR_STAGGER=1 CTCASR_LIB=$lib
  run CTCASR_STAGGER=1 CTCASR_LIB=$lib CTCASR_RNN_PROF=1
done
cat $out
