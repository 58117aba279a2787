R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
python $R/tools/dev_bench_decoder_graph.py 32
cd /tmp && rm -rf /tmp/gp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/gp -o t -- python $R/tools/dev_bench_decoder_graph.py 32 > /dev/null 2>&1
F=$(find /tmp/gp -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last replay: find the last point_tokens kernel and take everything from there
idx = [i for i, r in enumerate(rows) if "point_tokens" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
seg = rows[a:b]
t0 = int(seg[0]["Start_Timestamp"]); t1 = int(rows[b]["Start_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print("kernels per batch", len(seg), "span us", (t1 - t0) / 1e3, "busy us", busy / 1e3)
agg = collections.defaultdict(lambda: [0, 0.0])
prev_end = None
gaps = []
for r in seg:
    n = r["Kernel_Name"]
    import re
    m = re.search(r"(\w+_kernel)", n); n = m.group(1) if m else n[:40]
    agg[n][0] += 1; agg[n][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if prev_end is not None: gaps.append((int(r["Start_Timestamp"]) - prev_end) / 1e3)
    prev_end = int(r["End_Timestamp"])
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("  %-36s n=%3d %8.1f us" % (k, v[0], v[1]))
gaps.sort()
print("gaps: n", len(gaps), "sum", sum(gaps), "median", gaps[len(gaps)//2], "max", gaps[-1])
PY
