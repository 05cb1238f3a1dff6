#!/bin/bash
# Round 6: the whole GPU suite, smoke, then the default bench line
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err
tail -5 gpurun_out/r06_gputests.log; tail -2 gpurun_out/r06_smoke.log; python tools/show_bench.py gpurun_out/r06_bench.json 2>/dev/null | head -40
