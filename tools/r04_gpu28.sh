#!/bin/bash
# absmax with one atomic per workgroup, conv0 wrw16 reduce: tests, timings, C3 / C2, C3 kernel trace
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py -x -q -m gpu -k "conv0 or guard or range or absmax or deferred" 2>&1 | tail -3
python tools/conv0_microbench.py 32 999 2>&1 | grep -v amdgpu.ids
for i in 1 2; do
python bench.py --workload c3 --steps 12 --warmup 4 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r04_am_c3.json 2> gpurun_out/r04_am_c3.err
python tools/show_bench.py gpurun_out/r04_am_c3.json
python bench.py --workload c2 --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r04_am_c2.json 2> gpurun_out/r04_am_c2.err
python tools/show_bench.py gpurun_out/r04_am_c2.json
done
bash tools/kt_only.sh c3 kt_c3 > gpurun_out/kt_c3.log 2>&1
head -40 gpurun_out/kt_c3/kt.md | cut -c1-150
