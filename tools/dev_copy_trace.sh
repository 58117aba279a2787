#!/bin/bash
# developer: where do the __amd_rocclr_copyBuffer dispatches of a bench step come from?  kernel trace as CSV, then every
# copy with its grid size, duration and the kernels dispatched right before / after it (one image's worth).
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
cd /tmp && rm -rf /tmp/ctrace
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ctrace -o t -- python $R/bench.py --no-cpu-baseline --no-kernel-timer --steps 3 --warmup 2 --crowd-keep 0 > /dev/null 2>&1
F=$(find /tmp/ctrace -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"][:48]
n = len(rows)
cp = [i for i, r in enumerate(rows) if "copyBuffer" in r["Kernel_Name"]]
print("dispatches", n, "copies", len(cp))
agg = collections.defaultdict(lambda: [0, 0.0])
for i in cp:
    r = rows[i]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = (r.get("Grid_Size_X", r.get("Grid_Size", "?")), name(rows[i - 1]) if i else "-", name(rows[i + 1]) if i + 1 < n else "-")
    agg[key][0] += 1
    agg[key][1] += dur
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("grid %-10s n=%4d total %9.1f us avg %7.1f | prev %-48s next %s" % (k[0], v[0], v[1], v[1] / v[0], k[1], k[2]))
PY
