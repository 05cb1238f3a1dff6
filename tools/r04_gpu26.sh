#!/bin/bash
# wrw16: parity test, kernel timings, C3 / C2 steps with and without it
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "kernel_gradient_on_the_fp16 or conv_s12" 2>&1 | tail -5
CONV_MICROBENCH_OWN_ONLY=1 python tools/conv_microbench.py 32 500 2>&1 | grep -v amdgpu.ids
CONV_MICROBENCH_OWN_ONLY=1 python tools/conv_microbench.py 16 500 2>&1 | grep -v amdgpu.ids
for v in 1 0; do
  CTCASR_CONV_WRW_F16=$v python bench.py --workload c3 --steps 12 --warmup 4 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r04_wrw${v}_c3.json 2> gpurun_out/r04_wrw${v}_c3.err
  python tools/show_bench.py gpurun_out/r04_wrw${v}_c3.json
  CTCASR_CONV_WRW_F16=$v python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r04_wrw${v}_c2.json 2> gpurun_out/r04_wrw${v}_c2.err
  python tools/show_bench.py gpurun_out/r04_wrw${v}_c2.json
done
