#!/bin/bash
# round 3, session 2: GPU stream tests, the chains alone under rocprofv3, a short driver-like bench (no FCN / CPU legs)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3p}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_cc_gpu.py tests/test_stream1080p_gpu.py -x -q -m gpu -k "not fcn" > $O/gpu_tests.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -2 $O/gpu_tests.txt
[ $rc -eq 0 ] || { grep -E "Error|error|assert" $O/gpu_tests.txt | head -20; exit 1; }
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/p -o c -- python3 $GRAFT_REPO_ROOT/tools/chain_profile.py > $O/chain.log 2>&1 || { tail -5 $O/chain.log; exit 1; }
cd $GRAFT_REPO_ROOT
grep rep $O/chain.log
python3 tools/db_to_stats_csv.py $(find $O/p -name "*_results.db" | head -1) $O/chain_kernel_stats.csv
rm -rf $O/p
python3 - $O <<'PY'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1] + '/chain_kernel_stats.csv')))
for r in rows[:30]:
    print("%-34s calls %6s total %8.1f ms avg %8.1f us" % (r['Name'].split('(')[0][:34], r['Calls'], int(r['TotalDurationNs'])/1e6/2, float(r['AverageNs'])/1e3))
PY
timeout -k 10 500 python bench.py --gpus 1 --steps 10 --warmup 3 --fcn-frames 0 --cpu-frames 0 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - $O <<'PY'
import json, sys
d=json.load(open(sys.argv[1] + '/bench.json'))
print("value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"])
print("roofline", {k: d["roofline"][k] for k in ("frac","launch_ms","alone")})
PY
