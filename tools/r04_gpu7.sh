python bench.py --workload c5 --steps 24 --no-cpu-baseline --no-parity-probe > gpurun_out/r04_c5.json 2>gpurun_out/r04_c5.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_c5.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'))
print(json.dumps(d.get('other_workloads',{}).get('c5_from_disk'),indent=1))
PY
tail -5 gpurun_out/r04_c5.err
