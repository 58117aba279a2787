#!/bin/bash
# roctx ranges next to the kernel trace (run on the GPU box from the repo root): bash tools/prof_ranges.sh [frames]
# -> gpurun_out/roctx_ranges.txt: per range name, calls and host-side milliseconds (rocprofv3 --marker-trace), for a run of
#    tools/test.py --synthetic N --profile with the shipped configuration.
N=${1:-6}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/roctx_prof
rocprofv3 --marker-trace --kernel-trace --output-format csv -d /tmp/roctx_prof -- \
  python $R/tools/test.py -c $R/configs/crowdhuman_mi355x.yaml --synthetic $N --profile -s /tmp/roctx_res.json \
  environ.output_dir /tmp/roctx_out > /tmp/roctx_run.log 2>&1
python - <<PY > $R/gpurun_out/roctx_ranges.txt
import csv, glob, collections, json
rows = []
for f in glob.glob("/tmp/roctx_prof/**/*marker_api_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
agg = collections.OrderedDict()
for r in rows:
    name = r.get("Function") or r.get("Name") or r.get("Message") or "?"
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    a = agg.setdefault(name, [0, 0.0, []]); a[0] += 1; a[1] += d; a[2].append(d)
print("roctx ranges of tools/test.py --synthetic $N --profile (shipped EPS configuration; --profile synchronises the device at every stage end)")
print("(the first frames carry one-time costs: code-object loads, hipGraph captures -- read the median)")
print("%-28s %8s %12s %10s %10s" % ("range", "calls", "total ms", "median ms", "max ms"))
for k, (n, ms, ds) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    ds.sort()
    print("%-28s %8d %12.2f %10.3f %10.2f" % (k[:28], n, ms, ds[len(ds) // 2], ds[-1]))
try:
    print("timings_rank0.json:", json.dumps(json.load(open("/tmp/roctx_out/timings_rank0.json"))["stage_ms_per_image"]))
except Exception as e:
    print("no timings file:", e)
PY
tail -3 /tmp/roctx_run.log >> $R/gpurun_out/roctx_ranges.txt
