#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3n; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_cc_gpu.py tests/test_stream1080p_gpu.py -x -q -m gpu -k "not fcn" > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/gpu_tests.txt
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/p -o c -- python3 $GRAFT_REPO_ROOT/tools/chain_profile.py > $O/chain.log 2>&1 || { tail -5 $O/chain.log; exit 1; }
cd $GRAFT_REPO_ROOT
grep rep $O/chain.log
python3 tools/db_to_stats_csv.py $(find $O/p -name "*_results.db" | head -1) $O/chain_kernel_stats.csv
rm -rf $O/p
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r3n/chain_kernel_stats.csv')))
for r in rows[:24]:
    print("%-34s calls %6s total %8.1f ms avg %8.1f us" % (r['Name'].split('(')[0][:34], r['Calls'], int(r['TotalDurationNs'])/1e6/2, float(r['AverageNs'])/1e3))
PY
