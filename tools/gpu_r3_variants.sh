#!/bin/bash
# A/B of library builds (tools/variants/lib_<name>.so, "default" = the in-tree library) on 1,280 dense frames of the bench stream:
# per-kernel averages of the chains alone.   bash tools/gpu_r3_variants.sh <outdir> <name> [<name> ...]
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O
for v in "$@"; do
  if [ "$v" = default ]; then unset LM_VARIANT_LIB; else export LM_VARIANT_LIB=$GRAFT_REPO_ROOT/tools/variants/lib_$v.so; fi
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p_$v -o c -- python3 $GRAFT_REPO_ROOT/tools/chain_profile.py 1280 5000 > $O/chain_$v.log 2>&1 || { tail -5 $O/chain_$v.log; exit 1; }
  cd $GRAFT_REPO_ROOT
  python3 tools/db_to_stats_csv.py $(find $O/p_$v -name "*_results.db" | head -1) $O/stats_$v.csv
  rm -rf $O/p_$v
  echo "== $v: $(grep 'rep 1' $O/chain_$v.log)"
  python3 - $O/stats_$v.csv <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
keep=("lm_k_emit","lm_k_stats","lm_k_select","lm_k_mb_tempo","lm_k_mb_twin","lm_k_mb_nt","lm_k_band","lm_k_pack","lm_k_write","lm_k_mb_resolve","lm_k_seam","lm_k_flatten","lm_k_apply","lm_k_mb_join","lm_k_mb_eval","lm_k_stats_init","lm_k_batch","lm_k_mb_sources","lm_k_mb_finish","fillBuffer")
out=[]
for r in rows:
    n=r['Name'].split('(')[0].replace('void ','')
    if any(k in n for k in keep): out.append("%s %.1f" % (n.replace('lm_k_',''), float(r['AverageNs'])/1e3))
print("  " + " | ".join(out))
PY
done
