# Kernel trace + timeline only (no PMC passes): tools/kt_only.sh <workload> <out> [ENV=VALUE ...]
W=${1:-c3}; O=$GRAFT_REPO_ROOT/gpurun_out/${2:-kt_$W}; R=$GRAFT_REPO_ROOT; shift; shift
mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_$W
B="python $R/bench.py --workload $W --no-cpu-baseline --no-other-workloads"
env "$@" rocprofv3 --kernel-trace --stats -d /tmp/kt_$W/kt -o $W -- $B --steps 4 --warmup 2 > $O/kt_bench.json 2>/dev/null
DB=$(find /tmp/kt_$W/kt -name "*.db" | head -1)
python $R/tools/prof_summary.py $DB 45 3 > $O/kt.md
python $R/tools/prof_summary.py --timeline $DB > $O/timeline.txt
