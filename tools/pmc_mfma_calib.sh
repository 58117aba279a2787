#!/bin/bash
# what does SQ_VALU_MFMA_BUSY_CYCLES count?  The probe issues a known number of MFMAs per wave at
# 1 / 2 / 4 waves per SIMD; counters per dispatch -> gpurun_out/r04_mfma_counter_calib.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
OUT=$R/gpurun_out/r04_mfma_counter_calib.txt
cd /tmp
$R/tools/probe/mfma_counter_calib > $OUT 2>&1
for G in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_CYCLES"; do
  rm -rf /tmp/pmc_cal
  timeout 300 rocprofv3 --pmc $G --kernel-trace -d /tmp/pmc_cal -o p -- $R/tools/probe/mfma_counter_calib > /tmp/pmc_cal.log 2>&1
  DB=$(find /tmp/pmc_cal -name "*.db" | head -1)
  python - "$DB" >> $OUT <<'PY'
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
name_col = "kernel_name" if "kernel_name" in cols else "name"
rows = {}
for name, cn, val, disp in c.execute(f"select {name_col}, counter_name, value, dispatch_id from counters_collection"):
    m = re.search(r"(\w+)(<[^>]*>)?", name)
    rows.setdefault((disp, m.group(0)), {}).setdefault(cn, 0.0)
    rows[(disp, m.group(0))][cn] += float(val)
for (disp, name), d in sorted(rows.items()):
    print("dispatch %3d %-22s " % (disp, name) + "  ".join("%s=%.6g" % kv for kv in sorted(d.items())))
PY
done
cat $OUT
