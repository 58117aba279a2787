R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
cd /tmp
: > $R/gpurun_out/pmc_cc.txt
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAVES" \
         "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
  rm -rf /tmp/pmc_cc
  timeout 600 rocprofv3 --pmc $G --kernel-trace -d /tmp/pmc_cc -o p -- python $R/tools/dev_bench_regions_idx.py > /tmp/pmc_cc.log 2>&1
  DB=$(find /tmp/pmc_cc -name "*.db" | head -1)
  for C in $G; do
    python $R/tools/pmc_summary.py $DB $C | python -c "
import json,sys
d=json.load(sys.stdin)
for k,v in sorted(d['kernels'].items(), key=lambda kv:-kv[1]['sum'])[:3]:
    if 'cc2' in k: print('%-24s %-40s launches %4d per_launch %.4g' % (d['counter'], k[:40], v['launches'], v['per_launch']))
" >> $R/gpurun_out/pmc_cc.txt
  done
done
cat $R/gpurun_out/pmc_cc.txt
