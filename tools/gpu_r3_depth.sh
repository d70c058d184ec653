#!/bin/bash
# bench pipeline variants: "<depth>:<back prio>" ...
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O
for v in "$@"; do
  d=${v%%:*}; bp=${v#*:}
  LM_BENCH_PRIO=${WIDE_PRIO:-front} LM_BENCH_BACK_PRIO=$bp timeout -k 10 400 python bench.py --gpus 1 --steps 9 --warmup 3 --depth $d --fcn-frames 0 --cpu-frames 0 > $O/bench_$d_$bp.json 2> $O/bench_$d_$bp.err || { tail -20 $O/bench_$d_$bp.err; exit 1; }
  python3 - $O/bench_$d_$bp.json $v <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print(sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"], "frac", d["roofline"]["frac"], "launch_ms", d["roofline"]["launch_ms"], "alone", d["roofline"]["alone"]["frac"])
PY
done
