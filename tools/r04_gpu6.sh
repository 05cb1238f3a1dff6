python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_dp_rccl.py -x -q -s -k "guard or out_of_range or deferred or collective or nccl_world" 2>&1 | grep -v amdgpu.ids | tail -15
python -m pytest tests/test_gpu_model.py -x -q -k "benchmark_shape or fifty" 2>&1 | tail -5
python bench.py --workload c3 --steps 12 --warmup 4 --no-cpu-baseline --no-other-workloads --no-parity-probe --collective-stand-in > gpurun_out/r04_standin.json 2>gpurun_out/r04_standin.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_standin.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['host_enqueue_ms_per_step'], json.dumps(d.get('collective_stand_in')))
PY
tail -3 gpurun_out/r04_standin.err
