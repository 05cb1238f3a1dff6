#!/bin/bash
# after the coalesced wrw reduces: conv tests, full profiling passes of C3 and C2 (final build)
mkdir -p gpurun_out
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" 2>&1 | tail -3
CONV_MICROBENCH_OWN_ONLY=1 python tools/conv_microbench.py 32 500 2>&1 | grep -v amdgpu.ids | tail -2
bash tools/prof_round.sh c3 r04/prof_c3_final > gpurun_out/prof_c3_final.log 2>&1
bash tools/prof_round.sh c2 r04/prof_c2_final > gpurun_out/prof_c2_final.log 2>&1
python tools/show_bench.py gpurun_out/r04/prof_c3_final/bench.json
python tools/show_bench.py gpurun_out/r04/prof_c2_final/bench.json
