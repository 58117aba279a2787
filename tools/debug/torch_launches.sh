#!/bin/bash
# Developer: how many kernels that are NOT libcsam_hip's does one timed frame launch, and what do they cost?
# rocprofv3 kernel trace of the serial bench loop; frames are delimited by the upscaler's launches (one per frame); everything whose
# name is not one of the library's kernels (torch elementwise / index / reduce / scan kernels, rocBLAS, blit copies and fills) is
# "torch-origin".  Run on the GPU box:  bash tools/debug/torch_launches.sh > gpurun_out/torch_launches.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
cd /tmp && rm -rf /tmp/tl && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o t -- \
  python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-kernel-timer --serial > /tmp/tl.log 2>&1
F=$(find /tmp/tl -name "*kernel_trace.csv" | head -1)
python - "$F" "$R" <<'PY'
import collections, csv, re, subprocess, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the library's kernel names, from its symbol table
syms = subprocess.run(["bash", "-c", "strings %s/crowdsam_amd/libcsam_hip.so | grep -E '_kernel' | head -4000" % sys.argv[2]], capture_output=True, text=True).stdout
own = syms            # mangled names: a kernel is the library's when its base name occurs in one of them
ups = [i for i, r in enumerate(rows) if "upscale_stream_kernel" in r["Kernel_Name"]]
a, b = ups[3], ups[8]            # five timed frames (launch 0 = setup, 1-2 = warm-up)
acc = collections.defaultdict(lambda: [0, 0.0])
n_own, t_own = 0, 0.0
for r in rows[a:b]:
    n = r["Kernel_Name"]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    m = re.search(r"(\w+_kernel)", n)
    base = m.group(1) if m else None
    if base and base in own and not any(t in n for t in ("at::native", "rocprim", "rocclr", "hipcub", "Cijk", "rocblas")):
        n_own += 1; t_own += dur
        continue
    key = re.sub(r"<.*", "", n.split("(")[0])[-60:]
    acc[key][0] += 1; acc[key][1] += dur
print("per timed frame (5 frames, serial loop): library kernels %.1f launches, %.2f ms" % (n_own / 5, t_own / 5e3))
tot_n = sum(c for c, _ in acc.values()); tot_t = sum(t for _, t in acc.values())
print("torch-origin launches (elementwise / index / reduce / scan / rocBLAS / blit): %.1f per frame, %.1f us per frame" % (tot_n / 5, tot_t / 5))
for k, (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    print("  %6.1f launches %8.1f us  %s" % (c / 5, t / 5, k))
PY
