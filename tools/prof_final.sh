set -x
cd /tmp && export TMPDIR=/tmp
R=/root/repo; O=$R/gpurun_out/final; mkdir -p $O
python $R/bench.py --steps 10 --warmup 3 > $O/bench.json 2>$O/bench.err
rocprofv3 --kernel-trace --stats -d /tmp/pf/kt -o c2 -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/kt_bench.json 2>/dev/null
python $R/tools/prof_summary.py $(find /tmp/pf/kt -name "*.db" | head -1) 45 5 > $O/kt.md
python $R/tools/prof_summary.py --timeline $(find /tmp/pf/kt -name "*.db" | head -1) > $O/timeline.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf/fetch -o c2 -- python $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/prof_summary.py --pmc $(find /tmp/pf/fetch -name "*.db" | head -1) 10 > $O/fetch.md
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pf/write -o c2 -- python $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/prof_summary.py --pmc $(find /tmp/pf/write -name "*.db" | head -1) 10 > $O/write.md
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pf/sq -o c2 -- python $R/bench.py --steps 2 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
python $R/tools/prof_summary.py --pmc $(find /tmp/pf/sq -name "*.db" | head -1) 40 > $O/sq.md
ls -la $O
