#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3tl}; mkdir -p $O
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace -d $O/p -o b -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 6 --warmup 1 --fcn-frames 0 --cpu-frames 0 > $O/bench.log 2>&1 || { tail -5 $O/bench.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 tools/prof_timeline.py $(find $O/p -name "*_results.db" | head -1) | tee $O/timeline.txt
python3 tools/db_to_stats_csv.py $(find $O/p -name "*_results.db" | head -1) $O/bench_kernel_stats.csv
rm -rf $O/p
