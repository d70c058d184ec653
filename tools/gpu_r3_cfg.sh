#!/bin/bash
# bench configurations "<depth>:<LM_LABEL_PARTS>" ...; the first is run twice (lease warm-up)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O
first=1
for v in "$@"; do
  d=${v%%:*}; p=${v#*:}
  for rep in $(seq 1 $((first + 1))); do
  LM_LABEL_PARTS=$p timeout -k 10 400 python bench.py --gpus 1 --steps 9 --warmup 3 --depth $d --fcn-frames 0 --cpu-frames 0 > $O/bench_d${d}_p$p.json 2> $O/bench_d${d}_p$p.err || { tail -20 $O/bench_d${d}_p$p.err; exit 1; }
  done
  first=0
  python3 - $O/bench_d${d}_p$p.json $v <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print("depth:parts", sys.argv[2], "value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"], "frac", d["roofline"]["frac"], "launch_ms", d["roofline"]["launch_ms"], "alone", d["roofline"]["alone"]["frac"])
PY
done
