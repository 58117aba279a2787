#!/bin/bash
# Developer: per timed frame of the pipelined bench, what each HIP stream (queue) does and when -- start / end / kernel time of
# the frame's own stream (sweep + tail) and of the look-ahead stream (encoder chunk + decoder constants), relative to the
# frame's upscaler launch.   bash tools/dev_stream_timeline.sh [bench args]   -> gpurun_out/r06_stream_timeline.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
cd /tmp && rm -rf /tmp/stl && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/stl -o t -- \
  python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timer "$@" > /tmp/stl.log 2>&1
F=$(find /tmp/stl -name "*kernel_trace.csv" | head -1)
python - "$F" "$@" > $R/gpurun_out/r06_stream_timeline.txt <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
    r["n"] = m.group(1) if m else r["Kernel_Name"][:40]
rows.sort(key=lambda r: r["s"])
ups = [i for i, r in enumerate(rows) if r["n"] == "upscale_stream_kernel"]
print("bench args:", sys.argv[2:], " upscaler launches:", len(ups))
first = 8 + 3                       # rehearsal (8 frames) + warm-up (3)
for k in range(first, first + 8):
    if k + 1 >= len(ups):
        break
    a, b = ups[k], ups[k + 1]
    t0 = rows[a]["s"]
    seg = rows[a:b]
    print("frame %d: period %.2f ms" % (k - first, (rows[b]["s"] - t0) / 1e6))
    byq = collections.defaultdict(list)
    for r in seg:
        byq[r["Queue_Id"]].append(r)
    for q, rs in sorted(byq.items(), key=lambda kv: kv[1][0]["s"]):
        ksum = sum(r["e"] - r["s"] for r in rs)
        top = collections.Counter()
        for r in rs:
            top[r["n"]] += r["e"] - r["s"]
        # gaps > 0.3 ms inside this queue
        gaps = [(rs[i + 1]["s"] - rs[i]["e"], rs[i]["n"], rs[i + 1]["n"], (rs[i]["e"] - t0) / 1e6) for i in range(len(rs) - 1)
                if rs[i + 1]["s"] - rs[i]["e"] > 300000]
        print("   queue %s: %4d kernels  +%.2f .. +%.2f ms  kernel time %.2f ms   top: %s" % (
            q, len(rs), (rs[0]["s"] - t0) / 1e6, (rs[-1]["e"] - t0) / 1e6, ksum / 1e6,
            ", ".join("%s %.2f" % (n.replace("_kernel", ""), v / 1e6) for n, v in top.most_common(4))))
        for g, n0, n1, at in gaps[:6]:
            print("        gap %.2f ms at +%.2f after %s before %s" % (g / 1e6, at, n0, n1))
PY
cat $R/gpurun_out/r06_stream_timeline.txt
