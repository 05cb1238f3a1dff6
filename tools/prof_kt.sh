cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/final; mkdir -p $O
python $R/bench.py --steps 10 --warmup 3 > $O/bench.json 2>$O/bench.err
rocprofv3 --kernel-trace --stats -d /tmp/pf/kt -o c2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/kt_bench.json 2>/dev/null
python $R/tools/prof_summary.py $(find /tmp/pf/kt -name "*.db" | head -1) 45 5 > $O/kt.md
