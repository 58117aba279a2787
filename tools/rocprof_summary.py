"""Summarise a rocprofv3 --kernel-trace --stats results .db (rocpd sqlite) into a text table."""
import sqlite3
import sys
import re


def short(name):
    m = re.search(r"(\w+_kernel)", name)
    if m:
        return m.group(1) + ("<true>" if "ILb1" in name else "<false>" if "ILb0" in name else "")
    return name[:60]


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    lines = ["%-44s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct")]
    for name, calls, tot, avg, pct in rows[:40]:
        lines.append("%-44s %8d %14.1f %12.2f %6.2f%%" % (short(name), calls, tot / 1e3 if tot > 1e6 else tot, avg / 1e3 if tot > 1e6 else avg, pct))
    text = "\n".join(lines)
    print(text)
    if out:
        open(out, "w").write(text + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
