"""FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, csv output) of tools/label_microbench.py -> the traffic JSON
bench.py reads (profiles/r02_label_traffic_pmc.json).  usage: pmc_traffic.py fetch.csv write.csv out.json [frames per launch, 64]
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of wide coalesced reads -> lm_k_pack_rows's image
read is doubled; the narrow run-table reads of the other kernels are left as reported.  Units: KB per dispatch."""
import collections, csv, json, sys
KERNELS = ["lm_k_pack_rows", "lm_k_band", "lm_k_seam_union", "lm_k_flatten_flag", "lm_k_apply_labels", "lm_k_write_labels"]
WIDE_READ = "lm_k_pack_rows"      # the only kernel that streams wide coalesced reads (the uint8 image, 16 B per lane)

def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].split("(")[0]
        if name in KERNELS:
            acc[name].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}

NF = int(sys.argv[4]) if len(sys.argv) > 4 else 64
fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
traffic_kb = sum(write.get(k, 0.0) for k in KERNELS) + sum(fetch.get(k, 0.0) * (2.0 if k == WIDE_READ else 1.0) for k in KERNELS)
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 tools/label_microbench.py %d" % NF,
       "frames_per_launch": NF,
       "note": "KB per dispatch, averaged over the dispatches of each kernel; gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads "
               "(MI355X_MICROARCH.md HBM): lm_k_pack_rows's image read is doubled in traffic_bytes, the narrow run-table reads are left as reported",
       "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
       "traffic_bytes_per_launch": int(traffic_kb * 1024), "traffic_bytes_per_frame": int(traffic_kb * 1024 / NF),
       "algorithmic_bytes_per_launch": 5 * 1920 * 1080 * NF}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out)[:600])
