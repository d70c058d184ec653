#!/bin/bash
# FCN: parity tests, bench of the fcn workload, kernel stats
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/fcn; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_cc_gpu.py tests/test_dropin_gpu.py -x -q -m gpu -k "fcn or FCN" 2>&1 | tail -5 || exit 1
timeout -k 10 300 python bench.py --workload fcn --steps 20 > $O/r02_fcn_bench.json 2> $O/fcn_bench.err || { tail -5 $O/fcn_bench.err; exit 1; }
cat $O/r02_fcn_bench.json
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p_fcn -o f -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcn --steps 10 --no-fcn-oracle > $O/p_fcn.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/db_to_stats_csv.py $(find $O/p_fcn -name "*_results.db" | head -1) $O/r02_fcn_kernel_stats.csv
python3 tools/fcn_layers.py $(find $O/p_fcn -name "*_results.db" | head -1) > $O/r02_fcn_layers.txt
rm -rf $O/p_fcn
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/fcn/r02_fcn_kernel_stats.csv')))
tot=sum(int(r['TotalDurationNs']) for r in rows)
print("total ms",tot/1e6)
for r in rows[:12]:
    print("%-70s %6s %9.2f ms %9.1f us %5.1f%%"%(r['Name'][:70],r['Calls'],int(r['TotalDurationNs'])/1e6,float(r['AverageNs'])/1e3,float(r['Percentage'])))
PY
cat gpurun_out/fcn/r02_fcn_layers.txt
