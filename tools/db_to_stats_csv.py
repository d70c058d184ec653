"""rocprofv3 (ROCm 7.2 rocpd .db) -> the per-kernel summary CSV committed under profiles/ (Name,Calls,TotalDurationNs,AverageNs,Percentage)."""
import csv, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end - start), avg(end - start) from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows) or 1
w = csv.writer(open(sys.argv[2], "w", newline=""))
w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage"])
for r in rows:
    w.writerow([r[0], r[1], int(r[2]), round(r[3], 1), round(100.0 * r[2] / tot, 3)])
