#!/bin/bash
# Which D2D / D2H copies does a frame issue?  rocprofv3 kernel trace of the EPS-mode bench, copyBuffer dispatches grouped by
# grid size with the kernel that precedes each (bash tools/dev_copy_trace2.sh [bench args]) -> gpurun_out/copy_trace.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
cd /tmp && rm -rf /tmp/ct
timeout 600 rocprofv3 --kernel-trace -d /tmp/ct -o p -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer "$@" > /dev/null 2>&1
DB=$(find /tmp/ct -name "*.db" | head -1)
python - "$DB" > $R/gpurun_out/copy_trace.txt <<'PY'
import sqlite3, sys, re, collections
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, duration, grid_x, start from kernels order by start"))
last = max(i for i, r in enumerate(rows) if "sam_im2col" in r[0])       # the LAST frame only (steady state)
rows = rows[last:]
print("last frame: %d dispatches, %.2f ms of kernel time, span %.2f ms" % (len(rows), sum(r[1] for r in rows) / 1e6, (rows[-1][3] + rows[-1][1] - rows[0][3]) / 1e6))
acc = collections.defaultdict(lambda: [0, 0.0])
prev = "?"
def short(n):
    m = re.search(r"(\w+)(<[^>]*>)?\(", n)
    return (m.group(1) if m else n[:40])
for name, dur, gx, st in rows:
    if "copyBuffer" in name:
        a = acc[(gx, prev)]
        a[0] += 1; a[1] += dur / 1e3
    prev = short(name)
tot = sum(v[1] for v in acc.values()); n = sum(v[0] for v in acc.values())
print("copyBuffer dispatches %d, total %.1f us" % (n, tot))
for (gx, pv), (k, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
    print("grid_x=%-9d after %-32s calls=%4d avg_us=%7.1f total_us=%9.1f" % (gx, pv, k, t / k, t))
PY
cat $R/gpurun_out/copy_trace.txt
