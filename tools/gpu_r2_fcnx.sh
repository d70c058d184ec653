#!/bin/bash
# A/B of library variants (tools/variants/liblm_<name>.so) on the FCN forward pass, all inside ONE lease: per-kernel averages over 30 passes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/fcnx; mkdir -p $O
for v in default "$@" default; do
  if [ $v = default ]; then unset LM_LIB_PATH; else export LM_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/liblm_$v.so; fi
  cd /tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/p_$v -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py f16x3 30 > $O/$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  grep "ms/frame" $O/$v.log
  python3 tools/db_to_stats_csv.py $(find $O/p_$v -name "*_results.db" | head -1) $O/stats_$v.csv > /dev/null
  python3 - $O/stats_$v.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("   " + " | ".join("%s %.0f" % (r["Name"].split("(")[0].replace("void lm_k_","")[:28], float(r["AverageNs"])/1e3) for r in rows[:9]))
PY
  rm -rf $O/p_$v
done
