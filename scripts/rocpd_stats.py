"""Summarise a rocprofv3 rocpd sqlite database into a per-kernel stats table (CSV on stdout)."""
import re
import sqlite3
import sys
con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
total = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for r in rows:
    name = r[0].replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    print('"%s",%d,%d,%.0f,%d,%d,%.2f' % (name[:110], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total))
