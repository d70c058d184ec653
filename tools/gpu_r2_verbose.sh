#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/verbose; mkdir -p $O
LM_BENCH_VERBOSE=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --depth 1 --fcn-frames 0 --cpu-frames 0 > $O/d1.json 2> $O/d1.err; tail -25 $O/d1.err
echo ----- depth 2
LM_BENCH_VERBOSE=1 timeout -k 10 300 python bench.py --steps 4 --warmup 2 --fcn-frames 0 --cpu-frames 0 > $O/d2.json 2> $O/d2.err; tail -25 $O/d2.err
python - <<'PY'
import json
for f in ("d1","d2"):
    d=json.loads(open("gpurun_out/verbose/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["roofline"]["frac"])
PY
