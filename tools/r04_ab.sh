#!/bin/bash
# Round-4 A/Bs of choices the cheaper (fp16) GEMMs may have flipped: each line
# is `bench.py --workload c3` at the final build of round 3, same box, same run.
mkdir -p gpurun_out/r04_ab
B="python bench.py --workload c3 --steps 12 --warmup 4 --no-cpu-baseline --no-other-workloads --no-parity-probe"
run() { name=$1; shift; echo "== $name"; env "$@" $B ${EXTRA} 2>gpurun_out/r04_ab/$name.err > gpurun_out/r04_ab/$name.json; python tools/show_bench.py gpurun_out/r04_ab/$name.json || tail -3 gpurun_out/r04_ab/$name.err; }
run default A=1
EXTRA=--rnn-bwd-whole-chip run bwd_whole_chip A=1
run chunks2 CTCASR_BWD_CHUNKS=2
run chunks4 CTCASR_BWD_CHUNKS=4
run fwd_pipe32 CTCASR_FWD_PIPELINE_MAX_BATCH=32
run default_again A=1
