#!/bin/bash
# schedule comparison: lm_stream_run_logits schedule 1 (gated in C) vs 0 (free) vs the Python-driven gated loop
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/sched; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_cc_gpu.py -x -q -m gpu -k "run_logits" 2>&1 | tail -3 || exit 1
for s in gated free gated-py; do
  LM_BENCH_SCHEDULE=$s timeout -k 10 400 python bench.py --gpus 1 --steps 6 --warmup 2 --fcn-frames 0 --cpu-frames 0 > $O/r02_bench_sched_$s.json 2> $O/err_$s.txt || { echo "bench $s failed"; tail -5 $O/err_$s.txt; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/sched/r02_bench_sched_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"], "roof", d["roofline"]["frac"], d["roofline"].get("alone",{}).get("frac"), "parity", d["parity"]["match"])
PY
