python -m pytest tests/test_gpu_kernels.py -x -q -k "fp16_matrix_pipe" 2>&1 | tail -4
python -m pytest tests/test_gpu_model.py tests/test_gpu_split_gemm.py tests/test_gpu_c5.py -x -q 2>&1 | tail -4
B="python bench.py --workload c3 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-parity-probe"
for i in 1 2; do
CTCASR_RNN_FWD_PIECES=0 $B > gpurun_out/r04_np_$i.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_np_$i.json
$B > gpurun_out/r04_p_$i.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_p_$i.json
done
