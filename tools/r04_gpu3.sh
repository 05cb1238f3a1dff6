python -m pytest tests/test_gpu_kernels.py -x -q -k "fp16_matrix_pipe or f16_scales" 2>&1 | tail -25
echo "== B=32 fp32"; CTCASR_RNN_PROF=1 python tools/rnn_microbench.py 500 32 1024 | grep -E "bwd|wg0|checksum" | tail -4
echo "== B=32 f16"; CTCASR_F16=1 CTCASR_RNN_PROF=1 python tools/rnn_microbench.py 500 32 1024 | grep -E "fwd|bwd|wg0|checksum"
echo "== B=32 f16 whole chip"; CTCASR_FULL=1 CTCASR_F16=1 CTCASR_RNN_PROF=1 python tools/rnn_microbench.py 500 32 1024 | grep -E "bwd|wg0"| tail -3
echo "== B=16 f16"; CTCASR_F16=1 CTCASR_RNN_PROF=1 python tools/rnn_microbench.py 500 16 1024 | grep -E "bwd|wg0" | tail -3
echo "== B=16 fp32"; CTCASR_RNN_PROF=1 python tools/rnn_microbench.py 500 16 1024 | grep -E "bwd|wg0" | tail -3
