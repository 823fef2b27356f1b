"""Summarise a rocprofv3 rocpd sqlite result (``*_results.db``) into a per-kernel table.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db [--by-grid] > profiles/rNN_x.txt
"""

import sqlite3
import sys


def main():
    path = sys.argv[1]
    by_grid = "--by-grid" in sys.argv
    db = sqlite3.connect(path)
    key = "name, grid_x, grid_y, grid_z, workgroup_x" if by_grid else "name"
    rows = db.execute(
        f"select {key}, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        f"max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by {key} "
        "order by sum(duration) desc"
    ).fetchall()
    total = sum(r[-7] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace summary of {path}")
    print(f"{'calls':>7} {'total_us':>11} {'avg_us':>9} {'min_us':>9} {'max_us':>9} {'pct':>6} {'vgpr':>5} {'agpr':>5} {'lds':>6}  kernel")
    for r in rows:
        name = r[0][:110]
        extra = f" grid=({r[1]},{r[2]},{r[3]}) wg={r[4]}" if by_grid else ""
        c, tot, avg, mn, mx, vg, ag, lds = r[-8:]
        print(f"{c:7d} {tot/1e3:11.1f} {avg/1e3:9.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*tot/total:6.2f} {vg:5d} {ag:5d} {lds:6d}  {name}{extra}")


if __name__ == "__main__":
    main()
