"""rocprofv3 rocpd sqlite results (``*_results.db``) on the command line:

    python tools/rocpd.py stats x_results.db [--by-grid]       per-kernel table of a --kernel-trace run
    python tools/rocpd.py pmc x_results.db [kernel-substring]  counters of a --pmc run, mean per kernel
    python tools/rocpd.py steps x_results.db [MARK]            the last decode step: busy time, span, launch gaps
"""
import sqlite3
import sys


def stats(argv):
    path = argv[0]
    by_grid = "--by-grid" in argv
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


def pmc(argv):
    path = argv[0]
    sub = argv[1] if len(argv) > 1 else ""
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


def steps(argv):
    db = sqlite3.connect(argv[0])
    mark = argv[1] if len(argv) > 1 else "decode_advance_kernel"
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
    rows = db.execute(f"select name, {start}, {end} from kernels order by {start}").fetchall()
    idx = [i for i, r in enumerate(rows) if mark in r[0]]
    if len(idx) < 3:
        sys.exit(f"fewer than three {mark} launches")
    step = rows[idx[-3] + 1: idx[-2] + 1]
    busy = sum(e - s for _, s, e in step)
    span = step[-1][2] - step[0][1]
    print(f"kernels in the step: {len(step)}; busy {busy/1e3:.1f} us; span {span/1e3:.1f} us; idle {100*(span-busy)/span:.1f} %")
    gaps = sorted(((step[i + 1][1] - step[i][2], step[i][0][:50], step[i + 1][0][:50]) for i in range(len(step) - 1)), reverse=True)
    print("mean gap %.2f us; largest:" % ((span - busy) / 1e3 / max(1, len(step) - 1)))
    for g, x, y in gaps[:8]:
        print(f"  {g/1e3:7.2f} us  after {x}  before {y}")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "steps": steps}[sys.argv[1]](sys.argv[2:])
