#!/bin/bash
# Round 5: (direction, batch tile) per pair of XCDs in two-tile forward launches - A/B
mkdir -p gpurun_out; out=gpurun_out/pairs.log; : > $out
L=ctc_asr_amd/csrc/_obj/alt_rnn_persistent_pairs.so
for lib in "" $L "" $L; do
  echo "== lib=$lib" >> $out
  CTCASR_LIB=$lib CTCASR_F16=1 CTCASR_XCD=1 timeout 300 python tools/rnn_microbench.py 500 32 1024 2>&1 | grep "fwd\|checksum" >> $out
done
CTCASR_LIB=$L timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "fp16_matrix_pipe or pipelined or rnn_fwd" 2>&1 | tail -2 >> $out
for lib in "" $L "" $L; do
  echo "== c3 lib=$lib" >> $out
  CTCASR_LIB=$lib timeout 300 python bench.py --workload c3 --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads 2>/dev/null > /tmp/b.json; python tools/show_bench.py /tmp/b.json | head -1 | cut -c1-110 >> $out
done
cat $out
