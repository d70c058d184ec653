#!/bin/bash
# bench configurations "<wide prio: front|none>:<match prio>:<back prio>" ...; the first is run twice (lease warm-up)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O
first=1
for v in "$@"; do
  IFS=: read wp mp bp <<< "$v"
  for rep in $(seq 1 $((first + 1))); do
  LM_BENCH_PRIO=$wp LM_BENCH_MATCH_PRIO=$mp LM_BENCH_BACK_PRIO=$bp timeout -k 10 400 python bench.py --gpus 1 --steps 9 --warmup 3 --fcn-frames 0 --cpu-frames 0 > $O/bench_$v.json 2> $O/bench_$v.err || { tail -20 $O/bench_$v.err; exit 1; }
  done
  first=0
  python3 - $O/bench_$v.json $v <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print("wide:match:back", sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"], "frac", d["roofline"]["frac"], "launch_ms", d["roofline"]["launch_ms"], "alone", d["roofline"]["alone"]["frac"])
PY
done
