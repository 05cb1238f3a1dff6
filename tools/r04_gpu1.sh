bash tools/r04_ab.sh 2>&1 | tee gpurun_out/r04_ab.log
echo "== full-size parity probe"
python bench.py --workload c3 --steps 4 --warmup 2 --no-cpu-baseline --no-other-workloads > gpurun_out/r04_probe_full.json 2> gpurun_out/r04_probe_full.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_probe_full.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('parity_probe'),indent=1))
print(d['dtype'])
PY
tail -5 gpurun_out/r04_probe_full.err
