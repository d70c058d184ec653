"""MFMA utilisation of the FCN kernels from a rocprofv3 --pmc pass (csv):
util = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs * 256 CUs * GRBM_GUI_ACTIVE / 8)   [GRBM_GUI_ACTIVE is summed over the 8 XCDs]
usage: fcn_mfma_pmc.py counter_collection.csv out.json"""
import collections, csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
d = collections.OrderedDict()
for r in rows:
    d.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.OrderedDict()
for v in d.values():
    if "GRBM_GUI_ACTIVE" not in v or "SQ_VALU_MFMA_BUSY_CYCLES" not in v:
        continue
    name = v["name"].split("(")[0]
    a = agg.setdefault(name, {"dispatches": 0, "mfma_busy_cycles": 0.0, "gui_active": 0.0})
    a["dispatches"] += 1
    a["mfma_busy_cycles"] += v["SQ_VALU_MFMA_BUSY_CYCLES"]
    a["gui_active"] += v["GRBM_GUI_ACTIVE"]
tot_b = tot_g = 0.0
for name, a in agg.items():
    a["mfma_util"] = round(a["mfma_busy_cycles"] / (1024.0 * a["gui_active"] / 8.0), 4) if a["gui_active"] else 0.0
    if "mfma" in name or "lm_k_g2" in name:
        tot_b += a["mfma_busy_cycles"]
        tot_g += a["gui_active"]
out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -- python3 bench.py --workload fcn --steps 2 --warmup 1",
       "definition": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8 XCDs)",
       "conv_stack_mfma_util": round(tot_b / (1024.0 * tot_g / 8.0), 4) if tot_g else None, "kernels": agg}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps({k: v["mfma_util"] for k, v in agg.items()}), out["conv_stack_mfma_util"])
