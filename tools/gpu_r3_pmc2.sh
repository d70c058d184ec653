#!/bin/bash
# per-kernel SQ counters of the chains alone on 640 dense frames (two passes), plus the list of available counters
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3pmc}; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1 || true
grep -oE "\b(SQ|TCC|TCP|TA|TD|GRBM)_[A-Z0-9_]+" $O/avail.txt | sort -u > $O/avail_names.txt; wc -l $O/avail_names.txt
K="lm_k_emit lm_k_stats lm_k_mb_tempo lm_k_band lm_k_seam_union lm_k_flatten_flag lm_k_apply_labels lm_k_mb_join lm_k_mb_eval lm_k_mb_nt lm_k_render_frames"
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $O/a -- python3 $GRAFT_REPO_ROOT/tools/chain_profile.py 640 5000 > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --output-format csv -d $O/b -- python3 $GRAFT_REPO_ROOT/tools/chain_profile.py 640 5000 > $O/b.log 2>&1 || { tail -5 $O/b.log; }
cd $GRAFT_REPO_ROOT
python3 tools/kernel_pmc.py $(ls $O/a/*/*counter_collection.csv | head -1) $K | tee $O/kernel_pmc_a.txt
python3 - $(ls $O/b/*/*counter_collection.csv | head -1) $K <<'PY' | tee $O/kernel_pmc_b.txt
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1]))); want = sys.argv[2:]
d = collections.OrderedDict()
for r in rows:
    e = d.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0].replace("void ", "")})
    e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.OrderedDict()
for v in d.values():
    if not any(v["name"].startswith(w) for w in want): continue
    a = agg.setdefault(v["name"], collections.defaultdict(float)); a["n"] += 1
    for k, x in v.items():
        if k != "name": a[k] += x
for name, a in agg.items():
    n = a["n"]; w = a.get("SQ_WAVES", 0) or 1
    print("%-24s n %3d " % (name[:24], n) + " ".join("%s/wave %.1f" % (k.replace("SQ_", ""), a[k] / w) for k in sorted(a) if k not in ("n", "SQ_WAVES")))
PY
rm -rf $O/a $O/b
