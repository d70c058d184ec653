#!/bin/bash
# bench with LM_RUN_AHEAD values given as arguments; the first is run twice (lease warm-up)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O
first=1
for r in "$@"; do
  for rep in $(seq 1 $((first + 1))); do
  LM_RUN_AHEAD=$r timeout -k 10 400 python bench.py --gpus 1 --steps 8 --warmup 3 --fcn-frames 0 --cpu-frames 0 > $O/bench_r$r.json 2> $O/bench_r$r.err || { tail -20 $O/bench_r$r.err; exit 1; }
  done
  first=0
  python3 - $O/bench_r$r.json $r <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print("run-ahead", sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"], "frac", d["roofline"]["frac"], "launch_ms", d["roofline"]["launch_ms"], "alone", d["roofline"]["alone"]["frac"])
PY
done
