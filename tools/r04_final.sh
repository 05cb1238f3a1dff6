# the round's closing run: every GPU test, then the default bench line (what the driver runs)
python -m pytest tests -x -q -m gpu 2>&1 | tail -4
python bench.py > gpurun_out/r04_default_line.json 2> gpurun_out/r04_default_line.err; echo rc=$?
python tools/show_bench.py gpurun_out/r04_default_line.json; tail -3 gpurun_out/r04_default_line.err
python -c "import __graft_entry__ as g; g.smoke()"
