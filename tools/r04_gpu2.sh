python -m pytest tests/test_gpu_kernels.py -x -q -k "f16" 2>&1 | tail -15
for B in 32 16; do
python tools/rnn_fwd_f16_probe.py 500 $B 1024 lstm
done
python tools/rnn_fwd_f16_probe.py 500 16 2048 lstm
python tools/rnn_fwd_f16_probe.py 500 32 1024 gru
W_MULT=2 python tools/rnn_fwd_f16_probe.py 500 32 1024 lstm
echo "== phases fp32"; CTCASR_RNN_PROF=1 python tools/rnn_microbench.py 500 32 1024 | head -8
echo "== phases f16"; CTCASR_F16=1 CTCASR_RNN_PROF=1 python tools/rnn_microbench.py 500 32 1024 | head -8
echo "== phases f16 B16"; CTCASR_F16=1 CTCASR_RNN_PROF=1 python tools/rnn_microbench.py 500 16 1024 | head -8
B="python bench.py --workload c3 --steps 12 --warmup 4 --no-cpu-baseline --no-other-workloads --no-parity-probe"
$B > gpurun_out/r04_c3_f32rec.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_c3_f32rec.json
CTCASR_RNN_FWD_F16=1 $B > gpurun_out/r04_c3_f16rec.json 2>gpurun_out/r04_c3_f16rec.err; python tools/show_bench.py gpurun_out/r04_c3_f16rec.json || tail gpurun_out/r04_c3_f16rec.err
CTCASR_RNN_FWD_F16=1 python bench.py --workload c2 --steps 12 --warmup 4 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r04_c2_f16rec.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_c2_f16rec.json
