"""Timeline of the LAST decode step in a rocprofv3 kernel trace (rocpd sqlite): kernels between the last two launches of
MARK (default: decode_advance_kernel), their durations, the span, the idle gaps, and the largest gaps with their neighbours.

    python tools/rocpd_steps.py x_results.db [MARK]
"""
import sqlite3, sys

db = sqlite3.connect(sys.argv[1])
mark = sys.argv[2] if len(sys.argv) > 2 else "decode_advance_kernel"
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
start, end = ("start", "end") if "start" in cols else ("start_timestamp", "end_timestamp")
rows = db.execute(f"select name, {start}, {end} from kernels order by {start}").fetchall()
idx = [i for i, r in enumerate(rows) if mark in r[0]]
if len(idx) < 3:
    sys.exit(f"fewer than three {mark} launches")
a, b = idx[-3] + 1, idx[-2] + 1
step = rows[a:b]
busy = sum(e - s for _, s, e in step)
span = step[-1][2] - step[0][1]
print(f"kernels in the step: {len(step)}; busy {busy/1e3:.1f} us; span {span/1e3:.1f} us; idle {100*(span-busy)/span:.1f} %")
gaps = sorted(((step[i + 1][1] - step[i][2], step[i][0][:50], step[i + 1][0][:50]) for i in range(len(step) - 1)), reverse=True)
print("mean gap %.2f us; largest:" % ((span - busy) / 1e3 / max(1, len(step) - 1)))
for g, x, y in gaps[:8]:
    print(f"  {g/1e3:7.2f} us  after {x}  before {y}")
