#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "=== tests"; python -m pytest tests/test_sharded_gpu.py tests/test_dropin_gpu.py tests/test_cc_gpu.py -x -q -k "sharded or 4k or entry or rebuilt or g2 or tie or age_boundaries or step_05 or pipeline" 2>&1 | tail -5
echo "=== label microbench"
for v in "" tools/variants/lib_band16.so; do
  LM_LIB_PATH=$v python tools/label_microbench.py 64 1080 1920 5000 2>&1 | grep labels=True
  LM_LIB_PATH=$v python tools/label_microbench.py 16 2160 3840 600 2>&1 | grep labels=True
done
echo "=== 4K bench"
timeout -k 10 400 python bench.py --height 2160 --width 3840 --frames 1024 --batch 16 --steps 3 --warmup 1 --cpu-frames 40 --fcn-frames 0 > gpurun_out/h_bench_4k.json 2> gpurun_out/h_bench_4k.err; echo rc=$?; tail -c 600 gpurun_out/h_bench_4k.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/h_bench_4k.json").read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("alone"), d["config"]["stream"], d["cpu_baseline"])
PY
