#!/bin/bash
# Round 6: kernel trace + PMC passes of the given workloads (tools/prof_round.sh) -> gpurun_out/r06/prof_<w>
for w in "$@"; do timeout 1500 bash tools/prof_round.sh $w r06/prof_$w > gpurun_out/r06_prof_$w.log 2>&1; done
ls gpurun_out/r06/*
