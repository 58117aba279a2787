#!/bin/bash
# developer: which hipBLASLt kernels does torch.mm pick for the path's GEMM shapes (names encode macro-tile / depth)?
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
cd /tmp && rm -rf /tmp/vk
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/vk -o t -- python $R/tools/probe/gemm_lib_probe.py > /dev/null 2>&1
F=$(find /tmp/vk -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0, None])
for r in rows:
    n = r["Kernel_Name"]
    if "Cijk" in n or "gemm" in n.lower():
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = (n, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")), r.get("LDS_Block_Size", "?"), r.get("VGPR_Count", "?"), r.get("Accum_VGPR_Count", "?"))
        agg[k][0] += 1; agg[k][1] += d
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("n=%4d avg %7.1f us grid %s wg %s lds %s vgpr %s agpr %s | %s" % (v[0], v[1] / v[0], k[1], k[2], k[3], k[4], k[5], k[0][:230]))
PY
