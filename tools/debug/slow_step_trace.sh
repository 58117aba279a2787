#!/bin/bash
# Developer: which kernel instances are the slowest of a bench run?  (a frame-specific 70 ms stall was traced with this)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
cd /tmp && rm -rf /tmp/sst && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/sst -o t -- \
  python $R/bench.py --steps ${1:-12} --warmup 2 --no-cpu-baseline --no-kernel-timer --serial > /tmp/sst.log 2>&1
F=$(find /tmp/sst -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
top = sorted(rows, key=lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), reverse=True)[:14]
for r in top:
    m = re.search(r"(\w+_kernel\w*|\w+)", r["Kernel_Name"])
    print("%10.1f us  at %9.1f ms  grid %8s  %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
          (int(r["Start_Timestamp"]) - t0) / 1e6, r["Grid_Size_X"], r["Kernel_Name"][:70]))
print("largest idle gaps between consecutive kernels (after the first 3 s):")
ends = 0; gaps = []
for i, r in enumerate(rows):
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if ends and st - ends > 2e6 and st - t0 > 3e9:
        gaps.append((st - ends, st, rows[i - 1]["Kernel_Name"][:50], r["Kernel_Name"][:50]))
    ends = max(ends, en)
for g, st, a, b in sorted(gaps, reverse=True)[:8]:
    print("%8.1f ms gap before %9.1f ms: after %s -> %s" % (g / 1e6, (st - t0) / 1e6, a, b))
PY
