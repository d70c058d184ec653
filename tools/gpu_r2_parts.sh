#!/bin/bash
# lm_label_batch in parts on two queues: parity, then the microbench for 1, 2, 3, 4, 8 parts, then the bench
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/parts; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_cc_gpu.py -x -q -m gpu -k "in_parts or run_logits or label" 2>&1 | tail -3 || exit 1
for p in 1 2 3 4 8; do
  LM_LABEL_PARTS=$p timeout -k 10 200 python tools/label_microbench.py 64 1080 1920 5000 2>&1 | grep labels= | sed "s/^/parts=$p /" | tee -a $O/r02_label_parts.txt
done
LM_LABEL_PARTS=4 timeout -k 10 200 python tools/label_microbench.py 64 1080 1920 192 2>&1 | grep labels= | sed "s/^/parts=4 /" | tee -a $O/r02_label_parts.txt
for p in 1 2 4; do
  LM_LABEL_PARTS=$p timeout -k 10 400 python bench.py --gpus 1 --steps 6 --warmup 2 --fcn-frames 0 --cpu-frames 0 > $O/r02_bench_parts_$p.json 2> $O/err_$p.txt || { echo "bench $p failed"; tail -5 $O/err_$p.txt; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/parts/r02_bench_parts_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], d["value"], d["ms_per_step"], "roof", d["roofline"]["frac"], d["roofline"].get("alone",{}).get("frac"), "parity", d["parity"]["match"])
PY
