#!/bin/bash
# SQ issue counters of the attention kernels (tools/dev_bench_attn.py): bash tools/pmc_sq_attn.sh -> gpurun_out/pmc_sq_attn.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
: > $R/gpurun_out/pmc_sq_attn.txt
for G in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_MFMA"; do
  rm -rf /tmp/pmc_sqa
  timeout 600 rocprofv3 --pmc $G --kernel-trace -d /tmp/pmc_sqa -o p -- python $R/tools/dev_bench_attn.py > /tmp/pmc_sqa.log 2>&1
  DB=$(find /tmp/pmc_sqa -name "*.db" | head -1)
  for C in $G; do
    python $R/tools/pmc_summary.py $DB $C | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['sum'])[:4]:
    print('%-28s %-44s launches %4d per_launch %.4g' % (d['counter'], k[:44], v['launches'], v['per_launch']))
" >> $R/gpurun_out/pmc_sq_attn.txt
  done
done
