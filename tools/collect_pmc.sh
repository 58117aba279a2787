#!/bin/bash
# HBM traffic per kernel from the rocprofv3 PMC counters, as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE passes (they do not fit one TCC pass), --kernel-trace only, graphs off so every launch
# is a plain dispatch.  Run on the GPU box from the repo root:  bash tools/collect_pmc.sh [bench args]
# Output: gpurun_out/pmc_fetch.json, gpurun_out/pmc_write.json, gpurun_out/pmc_traffic.json
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R CSAM_GRAPHS=0 TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$C
  timeout 900 rocprofv3 --pmc $C --kernel-trace -d /tmp/pmc_$C -o p -- \
    python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-timer --no-extra-legs "$@" > /tmp/pmc_$C.log 2>&1
  DB=$(find /tmp/pmc_$C -name "*.db" | head -1)
  python $R/tools/pmc_summary.py $DB $C > $R/gpurun_out/pmc_$(echo $C | cut -d_ -f1 | tr A-Z a-z).json
done
python $R/tools/pmc_merge.py $R/gpurun_out/pmc_fetch.json $R/gpurun_out/pmc_write.json 3 > $R/gpurun_out/pmc_traffic.json
