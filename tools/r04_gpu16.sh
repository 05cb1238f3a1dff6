python -m pytest tests/test_gpu_kernels.py -x -q -k "fp16_matrix_pipe" 2>&1 | tail -2
echo "== alone, default mapping"; CTCASR_F16=1 python tools/rnn_microbench.py 500 32 1024 | grep -E "fwd:|bwd:|checksum"
echo "== alone, xcd split"; CTCASR_XCD=1 CTCASR_F16=1 python tools/rnn_microbench.py 500 32 1024 | grep -E "fwd:|bwd:|checksum"
echo "== B16"; CTCASR_F16=1 python tools/rnn_microbench.py 500 16 1024 | grep -E "fwd:|bwd:"; CTCASR_XCD=1 CTCASR_F16=1 python tools/rnn_microbench.py 500 16 1024 | grep -E "fwd:|bwd:"
B="python bench.py --workload c3 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-parity-probe"
for i in 1 2; do
$B > gpurun_out/r04_x0_$i.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_x0_$i.json
CTCASR_RNN_XCD_SPLIT=1 $B > gpurun_out/r04_x1_$i.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_x1_$i.json
done
CTCASR_RNN_XCD_SPLIT=1 python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r04_x1_c2.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_x1_c2.json
python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r04_x0_c2.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_x0_c2.json
