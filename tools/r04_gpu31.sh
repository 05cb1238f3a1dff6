#!/bin/bash
mkdir -p gpurun_out
bash tools/kt_only.sh c5 kt_c5 > gpurun_out/kt_c5.log 2>&1
python tools/show_bench.py gpurun_out/kt_c5/kt_bench.json
head -50 gpurun_out/kt_c5/kt.md | cut -c1-170
