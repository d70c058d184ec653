#!/bin/bash
# bench under the schedules given as arguments (gated | free), short runs without the FCN / CPU legs
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O
for s in "$@"; do
  LM_BENCH_SCHEDULE=$s timeout -k 10 400 python bench.py --gpus 1 --steps 8 --warmup 3 --fcn-frames 0 --cpu-frames 0 > $O/bench_$s.json 2> $O/bench_$s.err || { tail -20 $O/bench_$s.err; exit 1; }
  python3 - $O/bench_$s.json $s <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"], "frac", d["roofline"]["frac"], "launch_ms", d["roofline"]["launch_ms"], "alone", d["roofline"]["alone"]["frac"])
PY
done
