"""FETCH_SIZE / WRITE_SIZE (separate rocprofv3 --pmc passes, csv output) of tools/label_microbench.py -> the traffic JSON bench.py
reads (profiles/r03_label_traffic_pmc.json).  usage: pmc_traffic.py fetch.csv write.csv out.json [frames per launch, 64] [width height]
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE reports half of wide coalesced reads -> the image / logit reads of
lm_k_pack_rows and lm_k_pack_rows_logits (16 B per lane, streaming) are doubled; the narrow run-table reads of the other kernels are
left as reported.  Units: KB per dispatch.  Two launch sequences are summed up: the fused one (logits in) and the byte-frame one."""
import collections, csv, json, sys
MIDDLE = ["lm_k_band", "lm_k_seam_union", "lm_k_flatten_flag", "lm_k_apply_labels", "lm_k_write_labels"]
KERNELS = ["lm_k_pack_rows", "lm_k_pack_rows_logits"] + MIDDLE
WIDE_READ = ("lm_k_pack_rows", "lm_k_pack_rows_logits")

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
W, H = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (1920, 1080)
fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
def total(first):
    ks = [first] + MIDDLE
    return int(1024 * (sum(write.get(k, 0.0) for k in ks) + sum(fetch.get(k, 0.0) * (2.0 if k in WIDE_READ else 1.0) for k in ks)))
out = {"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) --output-format csv -- python3 tools/label_microbench.py %d %d %d" % (NF, H, W),
       "frames_per_launch": NF, "width": W, "height": H, "fused": True,
       "note": "KB per dispatch, averaged over the dispatches of each kernel; gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads "
               "(MI355X_MICROARCH.md HBM): the image / logit read of the packing kernel is doubled in the totals, the narrow run-table reads are left as reported",
       "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write,
       "hbm_bytes_per_launch": total("lm_k_pack_rows_logits"), "algorithmic_bytes_per_launch": 8 * W * H * NF,
       "byte_frame_launch": {"hbm_bytes_per_launch": total("lm_k_pack_rows"), "algorithmic_bytes_per_launch": 5 * W * H * NF}}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out)[:900])
