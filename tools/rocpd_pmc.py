"""Summarise the PMC counters of a rocprofv3 rocpd sqlite result per kernel (mean over dispatches).

    python tools/rocpd_pmc.py gpurun_out/pmc/x_results.db [kernel-substring]
"""

import sqlite3
import sys


def main():
    path = sys.argv[1]
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = db.execute(
        f"select {name_col}, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
        f"where {name_col} like ? group by {name_col}, counter_name order by {name_col}, counter_name",
        (f"%{sub}%",),
    ).fetchall()
    print(f"# rocprofv3 --pmc summary of {path}")
    last = None
    for name, ctr, n, avg, mn, mx in rows:
        if name != last:
            print(f"\n{name[:120]}")
            last = name
        print(f"   {ctr:32s} n={n:5d}  avg={avg:16.1f}  min={mn:16.1f}  max={mx:16.1f}")


if __name__ == "__main__":
    main()
