#!/bin/bash
# Developer: GPU timeline of the pipelined bench (rocprofv3 --kernel-trace csv): per steady-state image, how busy the GPU is,
# where it idles, and how the previous frame's tail kernels and the next frame's encoder kernels overlap.
#   bash tools/dev_pipeline_timeline.sh [--serial]   -> gpurun_out/r04_pipeline_timeline[_serial].txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R TMPDIR=/tmp
TAG=r05_pipeline_timeline$( [ "$1" == "--serial" ] && echo _serial )$( [ "$1" == "--batch" ] && echo _b$2 )
cd /tmp && rm -rf /tmp/ptl && timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/ptl -o t -- \
  python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timer --crowd-keep 720 "$@" > /tmp/ptl.log 2>&1
F=$(find /tmp/ptl -name "*kernel_trace.csv" | head -1)
python - "$F" > $R/gpurun_out/$TAG.txt <<'PY'
import csv, sys, re, collections
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    m = re.search(r"(\w+_kernel)", r["Kernel_Name"])
    r["n"] = m.group(1) if m else r["Kernel_Name"][:40]
rows.sort(key=lambda r: r["s"])
print("columns:", [k for k in rows[0].keys() if k not in ("s", "e", "n")])
qkey = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
# one image = from one upscale_stream_kernel start to the next (exactly one launch per image in the dense sweep)
ups = [i for i, r in enumerate(rows) if r["n"] == "upscale_stream_kernel"]
print("images traced:", len(ups))
TAIL = ("cc2_", "cc_", "rle_", "nms_", "mask_pack", "mask_cov")
# launches 0-7 = rehearsal of the pipelined loop, 8-10 = warm-up, 11-18 = the 8 timed frames, then the serial / collapsed legs
for k in range(11, 19):                                # the timed frames
    if k + 1 >= len(ups):
        break
    a, b = ups[k], ups[k + 1]
    seg = rows[a:b]
    t0, t1 = seg[0]["s"], rows[b]["s"]
    # union of busy intervals
    ev = sorted((r["s"], r["e"]) for r in seg)
    busy, cur_s, cur_e, gaps = 0, ev[0][0], ev[0][1], []
    for s, e in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    ksum = sum(r["e"] - r["s"] for r in seg)
    tail = [r for r in seg if r["n"].startswith(TAIL)]
    enc = [r for r in seg if r["n"] in ("gemm256_kernel", "gemm_f16_kernel", "flash_attn_kernel", "win_attn2_kernel")]
    print("image %d: period %.2f ms, GPU busy (union) %.2f ms, sum of kernel times %.2f ms (concurrency %.2f), %d kernels, "
          "idle %.2f ms in %d gaps (largest %s us)" % (k, (t1 - t0) / 1e6, busy / 1e6, ksum / 1e6, ksum / max(busy, 1), len(seg),
          (t1 - t0 - busy) / 1e6, len(gaps), [round(g[0] / 1e3) for g in sorted(gaps, reverse=True)[:5]]))
    if tail and enc:
        print("   tail kernels: %.2f ms of kernel time between +%.2f and +%.2f ms; encoder-side GEMM/attention kernels: %.2f ms "
              "between +%.2f and +%.2f ms" % (sum(r["e"] - r["s"] for r in tail) / 1e6, (tail[0]["s"] - t0) / 1e6, (tail[-1]["e"] - t0) / 1e6,
                                               sum(r["e"] - r["s"] for r in enc) / 1e6, (enc[0]["s"] - t0) / 1e6, (enc[-1]["e"] - t0) / 1e6))
    agg = collections.defaultdict(float)
    for r in seg:
        agg[r["n"]] += (r["e"] - r["s"]) / 1e6
    if k == 15:
        for n, v in sorted(agg.items(), key=lambda kv: -kv[1])[:22]:
            print("      %-36s %7.3f ms" % (n, v))
PY
cat $R/gpurun_out/$TAG.txt
