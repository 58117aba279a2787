#!/bin/bash
# developer: rocprofv3 kernel stats of one decoder batch size (tools/dev_bench_decoder.py <B>)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
cd /tmp && rm -rf /tmp/dp
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/dp -o p -- python $R/tools/dev_bench_decoder.py ${1:-2048} > /dev/null 2>&1
DB=$(find /tmp/dp -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/decoder_kernel_stats.txt > /dev/null
