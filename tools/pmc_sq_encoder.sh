#!/bin/bash
# SQ issue counters of the encoder-only bench (matrix-pipe busy cycles of the SAM encoder's kernels):
#   bash tools/pmc_sq_encoder.sh -> gpurun_out/pmc_sq_encoder.txt   (counters in their own passes, --kernel-trace only)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R CSAM_GRAPHS=0 TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
: > $R/gpurun_out/pmc_sq_encoder.txt
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_BUSY_CU_CYCLES"; do
  rm -rf /tmp/pmc_sqe
  timeout 600 rocprofv3 --pmc $G --kernel-trace -d /tmp/pmc_sqe -o p -- \
    python $R/bench.py --encoder-only --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-timer > /tmp/pmc_sqe.log 2>&1
  DB=$(find /tmp/pmc_sqe -name "*.db" | head -1)
  for C in $G; do
    python $R/tools/pmc_summary.py $DB $C | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['sum'])[:7]:
    print('%-28s %-44s launches %4d per_launch %.4g' % (d['counter'], k[:44], v['launches'], v['per_launch']))
" >> $R/gpurun_out/pmc_sq_encoder.txt
  done
done
