#!/bin/bash
# rocprofv3 --kernel-trace --stats of a bench.py command -> gpurun_out/<tag>_kernel_stats.txt (run on the GPU box)
#   bash tools/prof_bench.sh <tag> [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
export PYTHONPATH=$R TMPDIR=/tmp
mkdir -p $R/gpurun_out
cd /tmp
rm -rf /tmp/prof_$TAG
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o p -- python $R/bench.py --no-cpu-baseline --no-kernel-timer --no-extra-legs "$@" > $R/gpurun_out/${TAG}_bench.out 2>&1
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/${TAG}_kernel_stats.txt > /dev/null
grep '"metric"' $R/gpurun_out/${TAG}_bench.out > $R/gpurun_out/${TAG}_bench_line.json
