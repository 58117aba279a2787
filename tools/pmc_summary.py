"""Per-kernel sums of one PMC counter from a rocprofv3 --pmc rocpd .db  ->  JSON on stdout.

usage: pmc_summary.py results.db COUNTER [kernel-substring]
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB (derived from TCC_EA0_RDREQ/_WRREQ); the gfx950
correction of MI355X_MICROARCH.md (FETCH_SIZE reports half of a wide coalesced streaming read) is applied by
the caller, not here."""
import json
import re
import sqlite3
import sys


def main():
    db, counter = sys.argv[1], sys.argv[2]
    pat = sys.argv[3] if len(sys.argv) > 3 else ""
    c = sqlite3.connect(db)
    cols = [d[0] for d in c.execute("select * from counters_collection limit 1").description]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    acc = {}
    q = f"select {name_col}, counter_name, value, dispatch_id from counters_collection where counter_name = ?"
    for name, cn, val, disp in c.execute(q, (counter,)):
        if pat and pat not in name:
            continue
        m = re.search(r"(\w+)(<[^>]*>)?\(", name)
        key = (m.group(1) + (m.group(2) or "")) if m else name[:60]
        a = acc.setdefault(key, {"dispatches": set(), "sum": 0.0})
        a["dispatches"].add(disp)
        a["sum"] += float(val)
    out = {k: {"launches": len(v["dispatches"]), "sum": v["sum"], "per_launch": v["sum"] / max(1, len(v["dispatches"]))}
           for k, v in acc.items()}
    print(json.dumps({"counter": counter, "columns": cols, "kernels": out}))


if __name__ == "__main__":
    main()
