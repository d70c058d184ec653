"""Per-kernel counter averages from a rocprofv3 --pmc csv: python tools/kernel_pmc.py counter_collection.csv name1 name2 ..."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:]
d = collections.OrderedDict()
for r in rows:
    e = d.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0].replace("void ", ""), "grid": r.get("Grid_Size", ""), "wg": r.get("Workgroup_Size", "")})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.OrderedDict()
for v in d.values():
    if want and not any(v["name"].startswith(w) for w in want):
        continue
    a = agg.setdefault(v["name"], collections.defaultdict(float))
    a["n"] += 1
    for k, x in v.items():
        if k not in ("name", "grid", "wg"):
            a[k] += x
    a["grid"] = v["grid"]
for name, a in agg.items():
    n = a["n"]
    wc = a.get("SQ_WAVE_CYCLES", 0.0) / n or 1.0
    print("%-28s n %4d grid %s waves %8.0f waveMcyc %8.2f park%% %5.1f istall%% %5.1f active%% %5.1f valu/wave %7.0f salu/wave %6.0f vmem_rd/wave %6.1f lds/wave %6.1f" % (
        name[:28], n, a["grid"], a.get("SQ_WAVES", 0) / n, wc / 1e6, 100 * a.get("SQ_WAIT_ANY", 0) / n / wc, 100 * a.get("SQ_WAIT_INST_ANY", 0) / n / wc,
        100 * a.get("SQ_ACTIVE_INST_ANY", 0) / n / wc, a.get("SQ_INSTS_VALU", 0) / (a.get("SQ_WAVES", 0) or 1), a.get("SQ_INSTS_SALU", 0) / (a.get("SQ_WAVES", 0) or 1),
        a.get("SQ_INSTS_VMEM_RD", 0) / (a.get("SQ_WAVES", 0) or 1), a.get("SQ_INSTS_LDS", 0) / (a.get("SQ_WAVES", 0) or 1)))
