#!/bin/bash
# Round 6: per-workgroup phase timers of the LSTM-2048 backward kernel: plain, then K-split builds
#   tools/r06_kpair2048_prof.sh [alt build names of tools/build_alt_file.sh rnn_persistent ...]
mkdir -p gpurun_out; out=gpurun_out/r06_kpair2048_prof.log; : > $out
run() { echo "== $*" >> $out; env "$@" CTCASR_F16=1 CTCASR_RNN_PROF=1 timeout 300 python tools/rnn_microbench.py 500 16 2048 2>&1 | grep -A12 "^bwd" >> $out; }
run A=1
run CTCASR_KPAIR=1
for v in "$@"; do run CTCASR_KPAIR=1 CTCASR_LIB=ctc_asr_amd/csrc/_obj/alt_rnn_persistent_$v.so; done
cat $out
