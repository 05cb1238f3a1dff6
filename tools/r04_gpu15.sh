for w in ref_default ref_best c2_3conv c3_3conv; do
python bench.py --workload $w --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --parity-probe small > gpurun_out/r04_$w.json 2>gpurun_out/r04_$w.err; python tools/show_bench.py gpurun_out/r04_$w.json || tail -3 gpurun_out/r04_$w.err
done
