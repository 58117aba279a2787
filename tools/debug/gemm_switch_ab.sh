#!/bin/bash
# Developer: GEMM kernel time per frame (serial rocprofv3 trace of bench.py) under an environment switch, e.g.
#   bash tools/debug/gemm_switch_ab.sh CSAM_GEMM_96 0 1
R=${GRAFT_REPO_ROOT:-$(pwd)}
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v bash $R/tools/prof_bench.sh ab_$v --serial --steps 8 > /dev/null 2>&1
  python - "$R/gpurun_out/ab_${v}_kernel_stats.txt" "$VAR=$v" <<'PY'
import sys
g = tot = 0.0; n = 0
for l in open(sys.argv[1]):
    f = l.split()
    if len(f) >= 5 and f[1].isdigit():
        tot += float(f[2])
        if "gemm" in f[0]: g += float(f[2])
        if "upscale_stream" in f[0]: n = int(f[1])
print("%-22s GEMM %.2f ms / frame, all kernels %.2f ms / frame (%d frames)" % (sys.argv[2], g / n / 1e3, tot / n / 1e3, n))
PY
done
