# Prices the reduce-scatter form of the LSTM-1024 backward recurrence on the fp16 matrix pipe before
# building it: the fp32 reduce-scatter kernel with a quarter of its MFMAs
# (probe build, wrong results) has the exchange pattern, the barrier and the byte counts of an
# fp16 form and about its matrix-pipe time.  Alone and inside the C3 training step, next to the
# all-gather kernels.   tools/r05_rs_probe.sh > gpurun_out/r05_rs_probe.txt
L=ctc_asr_amd/csrc/_obj/alt_rsq1.so
echo "== alone, B = 32 (us per time step; phases of workgroup 0)"
echo "-- all-gather, fp16 pipe (the default)"; CTCASR_F16=1 CTCASR_XCD=1 CTCASR_RNN_PROF=1 python tools/rnn_microbench.py 500 32 1024 | grep -A1 "^bwd"
echo "-- reduce-scatter, fp32 MFMAs (round 3)"; CTCASR_RS=1 CTCASR_RNN_PROF=1 python tools/rnn_microbench.py 500 32 1024 | grep -A12 "^bwd" | head -14
echo "-- reduce-scatter, a quarter of the MFMAs (probe)"; CTCASR_ALLOW_PROBE_BUILD=1 CTCASR_LIB=$L CTCASR_RS=1 CTCASR_RNN_PROF=1 python tools/rnn_microbench.py 500 32 1024 | grep -A12 "^bwd" | head -14
echo "== alone, B = 16"
CTCASR_F16=1 CTCASR_XCD=1 python tools/rnn_microbench.py 500 16 1024 | grep "^bwd"
CTCASR_ALLOW_PROBE_BUILD=1 CTCASR_LIB=$L CTCASR_RS=1 python tools/rnn_microbench.py 500 16 1024 | grep "^bwd"
echo "== inside the C3 step (bench.py: ms per step, backward recurrence us per time step)"
line() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], 'ms;', d['roofline']['kernel'][:40], d['roofline']['us_per_time_step'], 'us per time step')"; }
B="python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-other-workloads --no-parity-probe"
$B 2>/dev/null | line "all-gather fp16 (default):"
CTCASR_RNN_BWD_F16=0 $B 2>/dev/null | line "all-gather fp32 MFMA:"
CTCASR_RNN_BWD_F16=0 CTCASR_RNN_BWD_FLAGS=8 $B 2>/dev/null | line "reduce-scatter fp32 MFMA:"
CTCASR_ALLOW_PROBE_BUILD=1 CTCASR_LIB=$L CTCASR_RNN_BWD_F16=0 CTCASR_RNN_BWD_FLAGS=8 $B 2>/dev/null | line "reduce-scatter, quarter of the MFMAs (probe):"
