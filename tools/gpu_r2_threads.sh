#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/thr; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_stream1080p_gpu.py tests/test_cc_gpu.py -x -q -m gpu -k "stream or grouping or 4k" 2>&1 | tail -3 || exit 1
LM_GROUP_TIMING=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --depth 1 --fcn-frames 0 --cpu-frames 0 > $O/d1.json 2> $O/d1.err; grep -E "entry lists|conflicts  |ages|split" $O/d1.err | tail -4
for t in 1 0; do
  if [ $t = 1 ]; then export LM_GROUP_THREADS=1; else unset LM_GROUP_THREADS; fi
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 --fcn-frames 0 --cpu-frames 0 > $O/b_$t.json 2>/dev/null
  python - $O/b_$t.json "threads=$t(1=single,0=auto)" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["parity"]["match"])
PY
done
unset LM_GROUP_THREADS
timeout -k 10 300 python bench.py --steps 6 --warmup 2 --fcn-frames 0 --cpu-frames 0 > $O/b_2.json 2>/dev/null
python - $O/b_2.json "auto again" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[2], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["parity"]["match"])
PY
