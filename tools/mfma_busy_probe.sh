# MFMA-busy and effective clock of every GEMM-like / convolution dispatch of a script:
#   tools/mfma_busy_probe.sh <out.md> python tools/split_gemm_kernel_probe.py
# (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace; busy = MFMA cycles /
#  (GRBM_GUI_ACTIVE / 8 XCDs x 256 CUs x 4 SIMDs), clock = GRBM_GUI_ACTIVE / 8 / duration)
out=$1; shift
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/mb
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/mb -o mb -- "$@" > /dev/null 2>&1
DB=$(find /tmp/mb -name "*.db" | head -1)
python - "$DB" > "$out" <<'P'
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
rows = con.execute("select kernel_name, counter_name, value, start, end from counters_collection "
                   "where counter_name in ('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE') "
                   "order by start").fetchall()
by = collections.OrderedDict()
for name, ctr, value, start, end in rows:
    by.setdefault((name, start, end), {})[ctr] = value
agg = collections.OrderedDict()
for (name, start, end), c in by.items():
    if 'GRBM_GUI_ACTIVE' not in c or 'SQ_VALU_MFMA_BUSY_CYCLES' not in c:
        continue
    dur = (end - start) / 1e3
    if dur < 60 or not any(tag in name for tag in ('Cijk', 'split_gemm', 'conv', 'wrw16')):
        continue
    key = name[:88]
    agg.setdefault(key, []).append((dur, c['GRBM_GUI_ACTIVE'] / 8.0 / (dur * 1e3),
                                    c['SQ_VALU_MFMA_BUSY_CYCLES'] /
                                    (c['GRBM_GUI_ACTIVE'] / 8.0 * 256 * 4)))
print('| kernel | dispatches | duration us (median) | effective clock GHz | MFMA-busy |')
print('|---|---:|---:|---:|---:|')
for key, v in agg.items():
    v.sort()
    d, clk, busy = v[len(v) // 2]
    print('| `{}` | {} | {:.0f} | {:.2f} | {:.2f} |'.format(key, len(v), d, clk, busy))
P
cat "$out"
