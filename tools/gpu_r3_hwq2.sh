#!/bin/bash
# bench with "<GPU_MAX_HW_QUEUES or default>:<depth>" ...; the first configuration is run twice (lease warm-up)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O
first=1
for v in "$@"; do
  q=${v%%:*}; d=${v#*:}
  if [ "$q" = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for rep in $(seq 1 $((first + 1))); do
  timeout -k 10 400 python bench.py --gpus 1 --steps 8 --warmup 3 --depth $d --fcn-frames 0 --cpu-frames 0 > $O/bench_q${q}_d$d.json 2> $O/bench_q${q}_d$d.err || { tail -20 $O/bench_q${q}_d$d.err; exit 1; }
  done
  first=0
  python3 - $O/bench_q${q}_d$d.json $v <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print("hwq:depth", sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"], "frac", d["roofline"]["frac"], "launch_ms", d["roofline"]["launch_ms"], "alone", d["roofline"]["alone"]["frac"])
PY
done
