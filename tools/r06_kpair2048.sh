#!/bin/bash
# Round 6: K pairs at H = 2048 (prnn_bwd16w_kernel<.., true>): parity test, then the kernel alone
mkdir -p gpurun_out; out=gpurun_out/r06_kpair2048.log; : > $out
[ -n "$SKIP_TEST" ] || timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "k_pairs_at_2048 or fp16_matrix_pipe" 2>&1 | tail -15 >> $out
run() { echo "== $*" >> $out; env "$@" CTCASR_ALLOW_PROBE_BUILD=1 CTCASR_F16=1 timeout 300 python tools/rnn_microbench.py 500 16 2048 2>&1 | grep "bwd\|phases\|checksum\|rror" >> $out; }
run A=1
run CTCASR_KPAIR=1
run CTCASR_RNN_PROF=1
run CTCASR_KPAIR=1 CTCASR_RNN_PROF=1
for v in "$@"; do
  lib=ctc_asr_amd/csrc/_obj/alt_rnn_persistent_$v.so
  run CTCASR_KPAIR=1 CTCASR_LIB=$lib
  run CTCASR_KPAIR=1 CTCASR_LIB=$lib CTCASR_RNN_PROF=1
done
cat $out
