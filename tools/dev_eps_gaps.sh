#!/bin/bash
# developer: GPU idle intervals over one image period of a bench mode (default: the shipped EPS configuration)
#   bash tools/dev_eps_gaps.sh [bench args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
ARGS=${@:---mode eps --grid 192 --points-per-batch 32 --stability-thresh 0.25}
cd /tmp && rm -rf /tmp/eg
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/eg -o t -- python $R/bench.py $ARGS --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --crowd-keep 0 > /dev/null 2>&1
F=$(find /tmp/eg -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"]); return m.group(1) if m else r["Kernel_Name"][:40]
idx = [i for i, r in enumerate(rows) if "sam_im2col" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"])
period = (int(rows[b]["Start_Timestamp"]) - t0) / 1e3
busy_union = 0.0
cur_end = int(seg[0]["Start_Timestamp"])
idle = []
for i, r in enumerate(seg):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > cur_end:
        idle.append(((s - cur_end) / 1e3, (cur_end - t0) / 1e3, nm(seg[i - 1]) if i else "-", nm(r)))
        busy_union += 0
    if e > cur_end:
        busy_union += (e - max(s, cur_end)) / 1e3
        cur_end = e
tail = (int(rows[b]["Start_Timestamp"]) - cur_end) / 1e3
print("image period %.1f us, GPU busy (union over streams) %.1f us, kernels %d, idle after the last kernel %.1f us" % (period, busy_union, len(seg), tail))
idle.sort(reverse=True)
print("idle intervals > 50 us: %d, total %.1f us" % (sum(1 for g in idle if g[0] > 50), sum(g[0] for g in idle if g[0] > 50)))
for g in idle[:25]:
    print("  idle %8.1f us at +%8.1f us  after %-34s before %s" % g)
small = [g[0] for g in idle if g[0] <= 50]
print("idle intervals <= 50 us: %d, total %.1f us" % (len(small), sum(small)))
PY
