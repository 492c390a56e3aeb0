"""Start / end of every dispatch of kernels whose name contains PATTERN, in start order, relative to the first one (us), with
the queue / stream columns the rocpd `kernels` view has.  usage: kernel_timeline_from_db.py results.db PATTERN [last_n]"""
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
extra = [c for c in ("queue_id", "stream_id", "grid_x", "grid_size_x", "grid_size", "workgroup_size") if c in cols]
rows = list(con.execute(f"select start, end{''.join(', ' + c for c in extra)} from kernels where name like ? order by start", (f"%{sys.argv[2]}%",)))
if len(sys.argv) > 3:
    rows = rows[-int(sys.argv[3]):]
t0 = rows[0][0]
print("columns: start_us end_us duration_us", *extra)
for r in rows:
    print(f"{(r[0] - t0) / 1e3:12.1f} {(r[1] - t0) / 1e3:12.1f} {(r[1] - r[0]) / 1e3:10.1f}", *r[2:])
