"""Per-launch stall picture of one FCN forward pass from a rocprofv3 --pmc pass (csv):
    SQ_WAVE_CYCLES ~ SQ_WAIT_ANY (parked at s_waitcnt / barrier) + SQ_WAIT_INST_ANY (issue stalls) + SQ_ACTIVE_INST_ANY   [quad-cycles]
usage: fcn_stall_pmc.py counter_collection.csv out.txt [launches_per_frame]"""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    e = d.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"], "grid": r.get("Grid_Size", ""), "lds": r.get("LDS_Block_Size", "")})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ds = [v for v in d.values() if v["name"].startswith(("lm_k", "void lm_k"))]
prep = [i for i, v in enumerate(ds) if v["name"].startswith("lm_k_prepare")]
if prep:
    ds = ds[prep[-1]:]          # the last forward pass
out = []
hdr = "%-46s %9s %6s %6s %6s %6s %6s %7s %7s" % ("kernel", "waveMcyc", "park%", "istal%", "activ%", "ldsst%", "mfma%", "ldsconf", "valu/wv")
out.append(hdr)
for v in ds:
    wc = v.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    name = v["name"].split("(")[0].replace("void ", "")
    mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    out.append("%-46s %9.2f %6.1f %6.1f %6.1f %6.1f %6.1f %7.3f %7.0f" % (
        name[:46], wc / 1e6, 100 * v.get("SQ_WAIT_ANY", 0) / wc, 100 * v.get("SQ_WAIT_INST_ANY", 0) / wc, 100 * v.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        100 * v.get("SQ_WAIT_INST_LDS", 0) / wc, 100 * (mf / 4.0) / wc if wc else 0,
        v.get("SQ_LDS_BANK_CONFLICT", 0) / (v.get("SQ_LDS_IDX_ACTIVE", 0) or 1.0), v.get("SQ_INSTS_VALU", 0) / (v.get("SQ_WAVES", 0) or 1.0)))
open(sys.argv[2], "w").write("\n".join(out) + "\n")
print("\n".join(out))
