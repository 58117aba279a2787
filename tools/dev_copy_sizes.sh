#!/bin/bash
# Developer: which device-side copies (__amd_rocclr_copyBuffer blit kernels) does one frame launch?  Grouped by grid size.
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
cd /tmp && rm -rf /tmp/cps && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/cps -o t -- \
  python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timer --serial > /tmp/cps.log 2>&1
F=$(find /tmp/cps -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ups = [i for i, r in enumerate(rows) if "upscale_stream_kernel" in r["Kernel_Name"]]
# timed crowded frames: upscaler launches 3..8 (0 = setup, 1-2 warm-up)
a, b = ups[3], ups[8]
seg = rows[a:b]
acc = collections.defaultdict(lambda: [0, 0.0])
prev = None
ctx = collections.defaultdict(collections.Counter)
for r in seg:
    n = r["Kernel_Name"]
    if "copyBuffer" in n or "fillBuffer" in n:
        key = (n.split("(")[0][-28:], int(r["Grid_Size_X"]), int(r["Workgroup_Size_X"]))
        acc[key][0] += 1
        acc[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        ctx[key][prev] += 1
    else:
        import re
        m = re.search(r"(\w+_kernel)", n); prev = m.group(1) if m else n[:30]
print("blit kernels per image over 5 timed frames:")
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("  %-30s grid %9d wg %4d: %6.1f calls/image %8.1f us/image  (%.1f us each)  after: %s"
          % (k[0], k[1], k[2], c / 5, t / 5, t / c, dict(ctx[k].most_common(2))))
print("total %.1f calls/image, %.1f us/image" % (sum(c for c, _ in acc.values()) / 5, sum(t for _, t in acc.values()) / 5))
PY
