"""Step intervals (ms) and GPU-busy time from a rocprofv3 rocpd database of `bench.py --no-pipeline`."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
th = [r[1] for r in rows if r[0].startswith("lm_k_threshold")]
print("step intervals ms:", [round((b - a) / 1e6, 2) for a, b in zip(th, th[1:])])
if len(th) >= 2:
    step = [r for r in rows if th[-2] <= r[1] < th[-1]]
    print("kernels/step", len(step), "busy ms", round(sum(r[2] - r[1] for r in step) / 1e6, 2))
    mb = [r for r in step if "lm_k_mb_" in r[0] or "lm_k_match" in r[0] or "lm_k_update" in r[0]]
    if mb:
        print("matching busy ms/step", round(sum(r[2] - r[1] for r in mb) / 1e6, 3), "span ms", round((mb[-1][2] - mb[0][1]) / 1e6, 2))
