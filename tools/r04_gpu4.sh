B="python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other-workloads --no-parity-probe"
$B --workload c3 > gpurun_out/r04_c3_f16both.json 2>gpurun_out/r04_c3_f16both.err; python tools/show_bench.py gpurun_out/r04_c3_f16both.json || tail gpurun_out/r04_c3_f16both.err
CTCASR_RNN_BWD_F16=0 $B --workload c3 > gpurun_out/r04_c3_f16fwd.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_c3_f16fwd.json
$B --workload c3 --rnn-bwd-whole-chip > gpurun_out/r04_c3_f16both_wc.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_c3_f16both_wc.json
CTCASR_BWD_CHUNKS=2 $B --workload c3 > gpurun_out/r04_c3_f16both_ch2.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_c3_f16both_ch2.json
$B --workload c2 > gpurun_out/r04_c2_f16both.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_c2_f16both.json
python -m pytest tests -x -q -m gpu 2>&1 | tail -15
