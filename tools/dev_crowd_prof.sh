R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
cd /tmp && rm -rf /tmp/cp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/cp -o p -- python $R/tools/dev_crowd_times.py > /dev/null 2>&1
DB=$(find /tmp/cp -name "*.db" | head -1)
python $R/tools/rocprof_summary.py $DB $R/gpurun_out/crowd_kernel_stats.txt > /dev/null
