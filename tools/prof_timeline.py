#!/usr/bin/env python3
"""Steady-state timeline of bench.py from a rocprofv3 --kernel-trace database: over the middle of the run, wall time per step,
time with >= 1 wide kernel running, idle time, time in which only single-workgroup kernels ran, and busy time per class of kernel
(labelling, records, matching, step 03) alone and overlapped.     python tools/prof_timeline.py results.db"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end,stream_id from kernels order by start").fetchall()
def cls(n):
    n = n.split("(")[0]
    if "rocclr" in n: return "blit"
    if any(k in n for k in ("pack_rows", "lm_k_band", "seam_union", "flatten_flag", "apply_labels", "write_labels", "lm_k_middle")): return "label"
    if any(k in n for k in ("lm_k_stats", "lm_k_select", "batch_offsets", "lm_k_emit")): return "records"
    if "lm_k_mb_resolve" in n or "lm_k_mb_sources" in n: return "replay"
    if "lm_k_mb_" in n: return "match"
    if "at::" in n: return "torch"
    return "step03"
packs = [r for r in rows if "pack_rows_logits" in r[0]]
# steps = runs of 157*2 pack launches; take the window from the start of the 3rd step's first pack to the start of the last step's first pack
per_step = int(sys.argv[2]) if len(sys.argv) > 2 else 157      # labelling launches of one step (157 batches x LM_LABEL_PARTS)
nsteps = len(packs) // per_step
a, b = 2, nsteps - 2          # the last step is the "alone" pass
t0, t1 = packs[a * per_step][1], packs[b * per_step][1]
R = [r for r in rows if r[2] > t0 and r[1] < t1]
def clip(s, e): return (max(s, t0), min(e, t1))
def union(iv):
    iv = sorted(iv)
    if not iv: return []
    out = []; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: out.append((cs, ce)); cs, ce = s, e
        else: ce = max(ce, e)
    out.append((cs, ce)); return out
def length(iv): return sum(e - s for s, e in iv)
def inter(x, y):
    i = j = 0; t = 0
    while i < len(x) and j < len(y):
        s = max(x[i][0], y[j][0]); e = min(x[i][1], y[j][1])
        if e > s: t += e - s
        if x[i][1] < y[j][1]: i += 1
        else: j += 1
    return t
n = b - a
ms = lambda x: x / 1e6 / n
U = {}
for c in ("label", "records", "match", "replay", "step03", "blit", "torch"):
    U[c] = union([clip(r[1], r[2]) for r in R if cls(r[0]) == c])
wide = union([clip(r[1], r[2]) for r in R if cls(r[0]) in ("label", "records", "match", "step03")])
anyk = union([clip(r[1], r[2]) for r in R if cls(r[0]) != "torch"])
print("steps %d: wall %.1f ms/step; a wide kernel running %.1f; only replay/blit running %.1f; idle %.1f" % (n, ms(t1 - t0), ms(length(wide)), ms(length(anyk) - length(wide)), ms((t1 - t0) - length(anyk))))
for c in ("label", "records", "match", "replay", "step03", "blit"):
    others = union([clip(r[1], r[2]) for r in R if cls(r[0]) not in (c, "torch", "blit", "replay")])
    print("  %-8s busy %6.1f ms/step, kernel sum %6.1f, of busy time %5.1f with another wide class running" % (c, ms(length(U[c])), ms(sum(min(r[2], t1) - max(r[1], t0) for r in R if cls(r[0]) == c)), ms(inter(U[c], others))))
for x, y in (("label", "match"), ("label", "step03"), ("label", "records"), ("records", "match"), ("records", "step03"), ("match", "step03")):
    print("  %s & %s together: %.1f ms/step" % (x, y, ms(inter(U[x], U[y]))))
