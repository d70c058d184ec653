"""HBM traffic of one FCN forward pass from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (csv) over tools/fcn_microbench.py:
per kernel (averaged over its dispatches) and per frame.  FETCH_SIZE on gfx950 reports half of wide coalesced reads (MI355X_MICROARCH.md, HBM
section); the LDS-DMA and 16-byte loads of these kernels are such reads, so the fetch figures are doubled (stated in the output).
usage: fcn_traffic_pmc.py fetch.csv write.csv out.json"""
import collections, csv, json, sys

def per_dispatch(path, counter):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            d.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"].split("(")[0].replace("void ", ""), 0.0])[1] += float(r["Counter_Value"])
    return list(d.values())

def last_frame(rows):
    idx = [i for i, (n, _) in enumerate(rows) if n.startswith("lm_k_prepare")]
    return rows[idx[-2]:idx[-1]] if len(idx) >= 2 else rows

f, w = last_frame(per_dispatch(sys.argv[1], "FETCH_SIZE")), last_frame(per_dispatch(sys.argv[2], "WRITE_SIZE"))
assert len(f) == len(w), (len(f), len(w))
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 tools/fcn_microbench.py mixed 4",
       "note": "KB per dispatch of ONE forward pass; FETCH_SIZE doubled (gfx950 reports half of wide coalesced reads)", "kernels": []}
tf = tw = 0.0
for (n, fv), (_, wv) in zip(f, w):
    out["kernels"].append({"kernel": n, "fetch_MB": round(2 * fv / 1024, 1), "write_MB": round(wv / 1024, 1)})
    tf += 2 * fv / 1024
    tw += wv / 1024
out["per_frame"] = {"fetch_MB": round(tf, 1), "write_MB": round(tw, 1), "total_MB": round(tf + tw, 1)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k in out["kernels"]:
    print("%-34s fetch %8.1f MB  write %8.1f MB" % (k["kernel"][:34], k["fetch_MB"], k["write_MB"]))
print("per frame:", out["per_frame"])
