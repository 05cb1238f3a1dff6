# Kernel trace + timeline + PMC passes of one bench workload, summarised under gpurun_out/<out>:
#   tools/prof_round.sh <workload> <out>
set -x
W=${1:-c3}; O=$GRAFT_REPO_ROOT/gpurun_out/${2:-r02/prof_$W}; R=$GRAFT_REPO_ROOT
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf_$W
B="python $R/bench.py --workload $W --no-cpu-baseline --no-other-workloads"
$B --steps 8 --warmup 3 > $O/bench.json 2>$O/bench.err
rocprofv3 --kernel-trace --stats -d /tmp/pf_$W/kt -o $W -- $B --steps 4 --warmup 2 > $O/kt_bench.json 2>/dev/null
DB=$(find /tmp/pf_$W/kt -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB 45 3 > $O/kt.md
python $R/tools/prof_summary.py --timeline $DB > $O/timeline.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf_$W/fetch -o $W -- $B --steps 2 --warmup 2 > /dev/null 2>&1
python $R/tools/prof_summary.py --pmc $(find /tmp/pf_$W/fetch -name "*.db" | head -1) 12 > $O/fetch.md
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pf_$W/write -o $W -- $B --steps 2 --warmup 2 > /dev/null 2>&1
python $R/tools/prof_summary.py --pmc $(find /tmp/pf_$W/write -name "*.db" | head -1) 12 > $O/write.md
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pf_$W/sq -o $W -- $B --steps 2 --warmup 2 > /dev/null 2>&1
python $R/tools/prof_summary.py --pmc $(find /tmp/pf_$W/sq -name "*.db" | head -1) 60 > $O/sq.md
ls -la $O
