# two-chain (8-wave) persistent kernels against the single-barrier kernels at B = 32 / 24
set -x
cd $GRAFT_REPO_ROOT
for B in 32 24; do
  for mode in "" "CTCASR_ONE_BARRIER=1"; do
    echo "== B=$B half-chip bwd $mode"; env $mode CTCASR_RNN_PROF=1 python tools/rnn_microbench.py 500 $B 1024 | grep -v "all 256\|busiest\|per blockIdx"
  done
done
echo "== B=32 whole-chip bwd (chains)"; CTCASR_FULL=1 python tools/rnn_microbench.py 500 32 1024 | grep -v "all 256\|busiest\|per blockIdx"
echo "== B=32 whole-chip bwd (one barrier)"; CTCASR_ONE_BARRIER=1 CTCASR_FULL=1 python tools/rnn_microbench.py 500 32 1024 | grep -v "all 256\|busiest\|per blockIdx"
echo "== relu 2048 B=32 chains"; python tools/rnn_microbench.py 500 32 2048 rnn_relu
echo "== relu 2048 B=32 one barrier"; CTCASR_ONE_BARRIER=1 python tools/rnn_microbench.py 500 32 2048 rnn_relu
python tools/gemm_microbench.py 16000 2048 8192
python tools/gemm_microbench.py 16000 640 8192
python tools/gemm_microbench.py 8000 2048 8192
