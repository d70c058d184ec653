#!/bin/bash
# round 3, job E: schedule / parts variants of the N = 1 bench + kernel stats of the default
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3e; mkdir -p $O
run() { # name, env...
  name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --gpus 1 --steps 8 --warmup 2 --fcn-frames 0 --cpu-frames 0 > $O/bench_$name.json 2> $O/bench_$name.err || { tail -5 $O/bench_$name.err; return 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3e/bench_$name.json')); r=d['roofline']
print('$name', 'value', d['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'launch_ms', r['launch_ms'], 'alone', r.get('alone',{}).get('frac'), 'parity', d['parity']['match'])"
}
run gated LM_BENCH_SCHEDULE=gated && run free LM_BENCH_SCHEDULE=free && run gated_parts1 LM_BENCH_SCHEDULE=gated LM_LABEL_PARTS=1 && run free_parts1 LM_BENCH_SCHEDULE=free LM_LABEL_PARTS=1 && run depth1 LM_BENCH_SCHEDULE=gated LM_X=1
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/p_bench -o b -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 5 --warmup 1 --fcn-frames 0 --cpu-frames 0 > $O/p_bench.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/db_to_stats_csv.py $(find $O/p_bench -name "*_results.db" | head -1) $O/bench_kernel_stats.csv
rm -rf $O/p_bench
head -40 $O/bench_kernel_stats.csv | cut -c1-60,200-
