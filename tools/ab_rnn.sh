# A/B of two library builds on the recurrence microbench: tools/ab_rnn.sh <alt.so> [B...]
ALT=$1; shift
for B in ${@:-16 32}; do
  for lib in "" "$ALT"; do
    for mode in "" "CTCASR_FULL=1"; do
      echo "== B=$B lib=${lib:-default} $mode"
      env CTCASR_LIB=$lib $mode python tools/rnn_microbench.py 500 $B 1024 | grep -v "busiest\|per blockIdx\|all 256\|amdgpu.ids"
    done
  done
done
