#!/bin/bash
# rocprofv3 kernel stats of one bench configuration: tools/gpu_r2_prof.sh <tag> <bench args...>
tag=$1; shift
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o $tag -- python3 bench.py "$@" > gpurun_out/prof_$tag/bench.json 2> gpurun_out/prof_$tag/bench.err
echo "rc=$?"
db=$(find gpurun_out/prof_$tag -name "*_results.db" | head -1)
python3 tools/db_to_stats_csv.py "$db" gpurun_out/prof_$tag/${tag}_kernel_stats.csv
f=gpurun_out/prof_$tag/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:-float(r["TotalDurationNs"]))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms %.1f"%(tot/1e6))
for r in rows[:45]:
    print("%-70s calls %7s avg_us %10.1f total_ms %9.2f  %5.1f%%"%(r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e6, 100*float(r["TotalDurationNs"])/tot))
PY
find gpurun_out/prof_$tag -name "*.db" -delete; find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
