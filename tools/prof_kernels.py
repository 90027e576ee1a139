#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run that wrote a rocpd sqlite database (the default output format).

  python tools/prof_kernels.py gpurun_out/prof_x [--csv out.csv]
"""
import glob
import sqlite3
import sys

d = sys.argv[1]
db = sorted(glob.glob(d + "/**/*.db", recursive=True))[0]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels "
                 "group by name order by 6 desc").fetchall()
lines = ["Name,Calls,AverageNs,MinNs,MaxNs,TotalNs"]
for r in rows:
    lines.append('"%s",%d,%.0f,%d,%d,%d' % (r[0], r[1], r[2], r[3], r[4], r[5]))
if "--csv" in sys.argv:
    open(sys.argv[sys.argv.index("--csv") + 1], "w").write("\n".join(lines) + "\n")
for r in rows[:30]:
    print("%-72s n=%6d avg=%9.1f us total=%9.2f ms" % (r[0][:72], r[1], r[2] / 1e3, r[5] / 1e6))
