B="python bench.py --workload c3 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-parity-probe"
for i in 1 2; do
CTCASR_BWD_CHUNKS=3 $B > gpurun_out/r04_ch3_$i.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_ch3_$i.json
CTCASR_BWD_CHUNKS=2 $B > gpurun_out/r04_ch2_$i.json 2>/dev/null; python tools/show_bench.py gpurun_out/r04_ch2_$i.json
done
python - <<'PY'
import json
for f in ('r04_ch3_1','r04_ch2_1'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1]); print(f, d['host_enqueue_ms_per_step'])
PY
