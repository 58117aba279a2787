"""Per-dispatch durations (us) from a rocprofv3 rocpd .db, grouped by (kernel, grid): avg over calls."""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
pat = sys.argv[2] if len(sys.argv) > 2 else ""
acc = {}
for name, dur, gx, gy, gz in c.execute("select name, duration, grid_x, grid_y, grid_z from kernels order by start"):
    if pat and pat not in name:
        continue
    m = re.search(r"(\w+)(<[^>]*>)?\(", name)
    key = ((m.group(1) + (m.group(2) or "")) if m else name[:50], gx, gy, gz)
    a = acc.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += dur / 1e3
for (k, gx, gy, gz), (n, t) in acc.items():
    print("%-40s grid=(%d,%d,%d) calls=%d avg_us=%.1f" % (k, gx, gy, gz, n, t / n))
if len(sys.argv) > 3:      # raw listing of the first N matching dispatches
    k = 0
    for name, dur, gy in c.execute("select name, duration, grid_y from kernels order by start"):
        if pat in name:
            m = re.search(r"(\w+)(<[^>]*>)?\(", name)
            print("%-30s gy=%d %.1f us" % (m.group(1) + (m.group(2) or ""), gy, dur / 1e3))
            k += 1
            if k >= int(sys.argv[3]):
                break
