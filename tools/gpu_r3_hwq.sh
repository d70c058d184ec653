#!/bin/bash
# bench with GPU_MAX_HW_QUEUES values given as arguments ("default" = unset); short runs
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O
for q in "$@"; do
  if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  timeout -k 10 400 python bench.py --gpus 1 --steps 8 --warmup 3 --fcn-frames 0 --cpu-frames 0 > $O/bench_q$q.json 2> $O/bench_q$q.err || { tail -20 $O/bench_q$q.err; exit 1; }
  python3 - $O/bench_q$q.json $q <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print("hwq", sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"], "frac", d["roofline"]["frac"], "launch_ms", d["roofline"]["launch_ms"], "alone", d["roofline"]["alone"]["frac"])
PY
done
