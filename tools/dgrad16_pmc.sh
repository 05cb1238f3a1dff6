# PMC passes over one kernel of a script (default: the block-scaled data-gradient kernel alone):
#   tools/dgrad16_pmc.sh <summary file under gpurun_out> [kernel substring] [command ...]
# Raw databases stay in /tmp; the summary lists, per counter, the mean over the kernel's dispatches.
out=gpurun_out/$1; shift
kern=${1:-dgrad16_bs}; shift
if [ $# -eq 0 ]; then set -- python tools/dgrad16_probe.py; fi
here=$(pwd); mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf /tmp/dgp; mkdir -p /tmp/dgp; : > $out
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_INSTS_SALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d /tmp/dgp/$i -o p -- "$@" > /tmp/dgp/$i.log 2>&1
  DB=$(find /tmp/dgp/$i -name "*.db" | head -1)
  python - "$DB" "$kern" >> $out <<'P'
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select kernel_name, counter_name, value, start, end from counters_collection "
                   "order by start").fetchall()
acc = collections.OrderedDict()
dur = []
seen = set()
for name, ctr, value, start, end in rows:
    if sys.argv[2] not in name:
        continue
    a = acc.setdefault(ctr, [0.0, 0]); a[0] += value; a[1] += 1
    if (start, end) not in seen:
        seen.add((start, end)); dur.append((end - start) / 1e3)
if dur:
    dur.sort()
    print('dispatches {} median duration {:.1f} us  '.format(len(dur), dur[len(dur) // 2]) +
          '  '.join('{} {:.4g}'.format(k, v[0] / v[1]) for k, v in acc.items()))
P
done
cat $out
