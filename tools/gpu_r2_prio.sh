#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/prio; mkdir -p $O
for p in front back none front; do
  LM_BENCH_PRIO=$p timeout -k 10 300 python bench.py --steps 6 --warmup 2 --fcn-frames 0 --cpu-frames 0 > $O/p_$p.json 2>/dev/null
  python - $O/p_$p.json $p <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["parity"]["match"])
PY
done
for dp in 3; do
  timeout -k 10 300 python bench.py --steps 6 --warmup 3 --depth $dp --fcn-frames 0 --cpu-frames 0 > $O/d_$dp.json 2>/dev/null
  python - $O/d_$dp.json depth$dp <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["parity"]["match"])
PY
done
