#!/usr/bin/env python3
"""GPU occupancy over the timed region of a rocprofv3 --kernel-trace database of bench.py: wall, sum of kernel durations, time
with at least one kernel running, per-stream busy time, and how much of the replay kernel's time something else ran."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,stream_id from kernels order by start").fetchall()
th = [r for r in rows if "threshold" in r[0]]
# bench.py: one step alone (reference digest) + 2 warm-up steps, then the timed steps, then one more step alone (un-shared timing)
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0 = th[skip][1]
last_alone = th[-1][1] if len(th) > skip + 1 else max(r[2] for r in rows) + 1
t_end = max(r[2] for r in rows if "lm_k_render_frames" in r[0] and r[1] < last_alone)
R = [r for r in rows if r[1] >= t0 and r[2] <= t_end]
def union(iv):
    iv = sorted(iv); out = []; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: out.append((cs, ce)); cs, ce = s, e
        else: ce = max(ce, e)
    out.append((cs, ce)); return out
def length(iv): return sum(e - s for s, e in iv)
def inter(a, b):
    i = j = 0; t = 0
    while i < len(a) and j < len(b):
        s = max(a[i][0], b[j][0]); e = min(a[i][1], b[j][1])
        if e > s: t += e - s
        if a[i][1] < b[j][1]: i += 1
        else: j += 1
    return t
t1 = max(r[2] for r in R)
steps = len([r for r in th if t0 <= r[1] < t_end])
allu = union([(r[1], r[2]) for r in R])
print("steps %d: wall %.2f ms/step, kernel sum %.2f, busy %.2f, idle %.2f" % (steps, (t1 - t0) / 1e6 / steps, sum(r[2] - r[1] for r in R) / 1e6 / steps,
      length(allu) / 1e6 / steps, ((t1 - t0) - length(allu)) / 1e6 / steps))
for sid in sorted(set(r[3] for r in R)):
    k = [r for r in R if r[3] == sid]
    print("  stream %d: %4d kernels, busy %.2f ms/step, e.g. %s" % (sid, len(k), length(union([(r[1], r[2]) for r in k])) / 1e6 / steps, k[len(k) // 2][0][:40]))
res = union([(r[1], r[2]) for r in R if "resolve" in r[0]])
oth = union([(r[1], r[2]) for r in R if "resolve" not in r[0]])
print("replay kernel: %.2f ms/step, of which %.2f with another kernel running" % (length(res) / 1e6 / steps, inter(res, oth) / 1e6 / steps))
for pat in ("lm_k_band", "lm_k_write_labels", "lm_k_threshold", "lm_k_mb_resolve", "lm_k_render_frames", "lm_k_stats", "lm_k_emit"):
    k = [r[2] - r[1] for r in R if pat in r[0]]
    if k: print("  %-22s avg %.1f us x %d" % (pat, sum(k) / len(k) / 1e3, len(k)))
