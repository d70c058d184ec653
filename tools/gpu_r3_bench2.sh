#!/bin/bash
# the driver's bench invocation twice on one lease (the first process on a fresh box runs slower), short form without FCN / CPU legs
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r3b2}; mkdir -p $O
for rep in 1 2 3; do
  timeout -k 10 400 python bench.py --gpus 1 --steps 10 --warmup 3 --fcn-frames 0 --cpu-frames 0 > $O/bench_$rep.json 2> $O/bench_$rep.err || { tail -20 $O/bench_$rep.err; exit 1; }
  python3 - $O/bench_$rep.json $rep <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print("rep", sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"], "frac", d["roofline"]["frac"], "launch_ms", d["roofline"]["launch_ms"], "alone", d["roofline"]["alone"]["frac"])
PY
done
