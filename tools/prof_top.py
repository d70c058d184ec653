import sqlite3, sys
db=sqlite3.connect(sys.argv[1])
n=int(sys.argv[2]) if len(sys.argv)>2 else 30
for r in db.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit %d"%n):
    print(f"{r[0][:58]:58s} calls={r[1]:6d} avg_us={r[3]/1e3 if r[3]>1e4 else r[3]:9.1f} tot={r[2]:10.1f} {r[4]:5.1f}%")
