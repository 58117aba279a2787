#!/bin/bash
# Developer: the kernels of ONE shipped-configuration EPS batch (32 prompts) in launch order, with durations and the idle gap in front
# of each -- what the sequential 16-batch chain is made of.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
cd /tmp && rm -rf /tmp/ebt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ebt -o t -- python $R/bench.py --mode eps --grid 192 --points-per-batch 32 \
  --stability-thresh 0.25 --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer --crowd-keep 0 --serial > /tmp/ebt.log 2>&1
F=$(find /tmp/ebt -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def nm(r):
    m = re.search(r"(\w+_kernel\w*)", r["Kernel_Name"]); return (m.group(1) if m else r["Kernel_Name"])[:44]
sel = [i for i, r in enumerate(rows) if "eps_select_kernel" in r["Kernel_Name"]]
a, b = sel[-6], sel[-5]                      # one batch in the middle of the last frame's chain
seg = rows[a:b]
prev_end = int(rows[a - 1]["End_Timestamp"])
tot_k = tot_g = 0.0
agg = collections.OrderedDict()
for r in seg:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - prev_end) / 1e3
    d = (e - s) / 1e3
    print("  +%6.1f gap  %7.1f us  grid %7s x %4s  %s" % (gap, d, r["Grid_Size_X"], r["Workgroup_Size_X"], nm(r)))
    tot_k += d; tot_g += gap; prev_end = max(prev_end, e)
    agg.setdefault(nm(r), [0, 0.0]); agg[nm(r)][0] += 1; agg[nm(r)][1] += d
period = (int(rows[b]["Start_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3
print("batch period %.1f us: %d kernels, kernel time %.1f us, idle gaps %.1f us" % (period, len(seg), tot_k, tot_g))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-44s x%2d %7.1f us" % (k, c, t))
PY
