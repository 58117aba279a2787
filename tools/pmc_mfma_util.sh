#!/bin/bash
# Counter-derived MFMA utilisation per kernel, with the calibration of tools/pmc_mfma_calib.sh:
# SQ_VALU_MFMA_BUSY_CYCLES charges 16 per v_mfma_f32_16x16x32_f16 (= its matrix-pipe occupancy), summed over all SIMDs;
# GRBM_GUI_ACTIVE is summed over the 8 XCDs.  utilisation = MFMA_BUSY / (1024 SIMDs x GUI_ACTIVE / 8).
#   bash tools/pmc_mfma_util.sh encoder   (bench.py --encoder-only)     -> gpurun_out/r06_mfma_util_encoder.txt
#   bash tools/pmc_mfma_util.sh frame     (bench.py, one crowded frame) -> gpurun_out/r06_mfma_util_frame.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R CSAM_GRAPHS=0 TMPDIR=/tmp
WHAT=${1:-encoder}
ARGS="--encoder-only --steps 3 --warmup 1"
[ "$WHAT" == "frame" ] && ARGS="--steps 2 --warmup 1 --serial"
[ "$WHAT" == "encoder4" ] && ARGS="--encoder-only --batch 4 --steps 2 --warmup 1"
mkdir -p $R/gpurun_out
cd /tmp && rm -rf /tmp/pmc_mu
timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU --kernel-trace -d /tmp/pmc_mu -o p -- \
  python $R/bench.py $ARGS --no-cpu-baseline --no-kernel-timer > /tmp/pmc_mu.log 2>&1
DB=$(find /tmp/pmc_mu -name "*.db" | head -1)
python - "$DB" > $R/gpurun_out/r06_mfma_util_$WHAT.txt <<'PY'
import sqlite3, sys, re, collections
c = sqlite3.connect(sys.argv[1])
cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
name_col = "kernel_name" if "kernel_name" in cols else "name"
per = collections.defaultdict(lambda: collections.defaultdict(float))
launch = collections.defaultdict(set)
for name, cn, val, disp in c.execute(f"select {name_col}, counter_name, value, dispatch_id from counters_collection"):
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
    k = (m.group(1) + (m.group(2) or "")) if m else name[:48]
    per[k][cn] += float(val)
    launch[k].add(disp)
print("MFMA utilisation from counters (calibration: profiles/r04_mfma_counter_calib.txt): SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)")
print("%-44s %7s %14s %14s %8s %10s" % ("kernel", "calls", "MFMA busy cyc", "active cyc/XCD", "util", "MFMA/VALU"))
tb = ta = 0.0
for k, d in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
    act = d.get("GRBM_GUI_ACTIVE", 0) / 8.0
    busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    if act <= 0:
        continue
    tb += busy; ta += act
    print("%-44s %7d %14.4g %14.4g %8.3f %10.3f" % (k[:44], len(launch[k]), busy, act, busy / (1024.0 * act),
                                                   d.get("SQ_INSTS_MFMA", 0) / max(d.get("SQ_INSTS_VALU", 1), 1)))
print("%-44s %7s %14.4g %14.4g %8.3f" % ("ALL KERNELS (time-weighted)", "", tb, ta, tb / (1024.0 * ta)))
own = {k: d for k, d in per.items() if not re.search(r"rocclr|at::native|rocblas|^elementwise_kernel|^reduce_kernel|vectorized_elementwise|index_elementwise", k)}
ob = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for d in own.values())
oa = sum(d.get("GRBM_GUI_ACTIVE", 0) / 8.0 for d in own.values())
print("%-44s %7s %14.4g %14.4g %8.3f   (without torch's copy / elementwise / plan-build kernels)" % ("libcsam_hip KERNELS (time-weighted)", "", ob, oa, ob / (1024.0 * oa)))
PY
cat $R/gpurun_out/r06_mfma_util_$WHAT.txt
