#!/bin/bash
# conv0 on the fp16 pipe: parity tests, kernel timings, model tests, C3 / C2 steps with and without
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv0" 2>&1 | tail -5
python tools/conv0_microbench.py 32 999 2>&1 | grep -v amdgpu.ids
python tools/conv0_microbench.py 16 999 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_model.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -4
for v in 1 0; do
  CTCASR_CONV_F16=$v python bench.py --workload c3 --steps 12 --warmup 4 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r04_cf${v}_c3.json 2> gpurun_out/r04_cf${v}_c3.err
  python tools/show_bench.py gpurun_out/r04_cf${v}_c3.json
  CTCASR_CONV_F16=$v python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r04_cf${v}_c2.json 2> gpurun_out/r04_cf${v}_c2.err
  python tools/show_bench.py gpurun_out/r04_cf${v}_c2.json
done
