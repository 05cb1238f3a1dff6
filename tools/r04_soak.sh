# soak runs of the round-4 build (deferred checks on): no drift, no time-out word, no hang
python bench.py --workload c3 --steps 400 --warmup 5 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r04_soak_c3.json 2>gpurun_out/r04_soak_c3.err; echo rc=$?; python tools/show_bench.py gpurun_out/r04_soak_c3.json
python bench.py --workload c2 --steps 1000 --warmup 5 --no-cpu-baseline --no-other-workloads --no-parity-probe > gpurun_out/r04_soak_c2.json 2>gpurun_out/r04_soak_c2.err; echo rc=$?; python tools/show_bench.py gpurun_out/r04_soak_c2.json
python bench.py --workload c5 --steps 96 --no-cpu-baseline --no-parity-probe > gpurun_out/r04_soak_c5.json 2>gpurun_out/r04_soak_c5.err; echo rc=$?; python tools/show_bench.py gpurun_out/r04_soak_c5.json
python tools/mixed_length_smoke.py 1280 16 3 20 2>&1 | tail -4
python -c "import __graft_entry__ as g; g.smoke()"
