"""Print a window of consecutive kernel dispatches (by start time) of a rocprofv3 rocpd database: which launches sit between which.
usage: python tools/prof_sequence.py DB [first_index] [count]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 120
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
t0 = "start" if "start" in cols else cols[0]
rows = list(db.execute(f"select name, grid_x, duration, {t0} from kernels order by {t0}"))
print(f"# {len(rows)} dispatches; columns of `kernels`: {cols}")
prev_end = None
for i, (name, grid, dur, st) in enumerate(rows[first:first + count]):
    gap = "" if prev_end is None else f"{(st - prev_end) / 1e3:8.2f}"
    print(f"{first + i:6d} {name[:70]:<70} {grid:>9} {dur / 1e3:9.2f} us  gap {gap}")
    prev_end = st + dur
