#!/bin/bash
# SQ issue/stall counters per kernel for one decoder batch: bash tools/pmc_sq.sh [B]  -> gpurun_out/pmc_sq.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R CSAM_GRAPHS=0 TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
B=${1:-2048}
: > $R/gpurun_out/pmc_sq.txt
for G in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM SQ_WAVES"; do
  rm -rf /tmp/pmc_sq
  timeout 600 rocprofv3 --pmc $G --kernel-trace -d /tmp/pmc_sq -o p -- python $R/tools/dev_bench_decoder.py $B > /tmp/pmc_sq.log 2>&1
  DB=$(find /tmp/pmc_sq -name "*.db" | head -1)
  for C in $G; do
    python $R/tools/pmc_summary.py $DB $C | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['sum'])[:8]:
    print('%-28s %-36s launches %4d per_launch %.4g' % (d['counter'], k[:36], v['launches'], v['per_launch']))
" >> $R/gpurun_out/pmc_sq.txt
  done
done
