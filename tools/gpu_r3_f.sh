#!/bin/bash
# round 3, job F: drop-in GPU tests (device resize, 4K frames), the configs[4] bench line
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3f; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_dropin_gpu.py -x -q -m gpu > $O/gpu_tests_dropin.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/gpu_tests_dropin.txt
timeout -k 10 900 python bench.py --gpus 1 --height 2160 --width 3840 --frames 1024 --batch 16 --steps 3 --warmup 1 --cpu-frames 0 > $O/bench_4k.json 2> $O/bench_4k.err || { tail -20 $O/bench_4k.err; exit 1; }
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3f/bench_4k.json'))
print("4K value", d["value"], "ms/step", d["ms_per_step"])
print("roofline", {k: d["roofline"][k] for k in ("frac","frac_survey_5Bpx","launch_ms","alone")})
print("fcn", {k: d["fcn"].get(k) for k in ("workload","precision","ms_per_frame","max_abs_logit_diff_vs_oracle")})
print("e2e", d.get("e2e_rgb"))
PY
