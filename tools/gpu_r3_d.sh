#!/bin/bash
# round 3, job D: the GPU suite, the driver's bench invocation, the two-rank rehearsal of the sharded bench on one GPU
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3d; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/gpu_tests.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3d/bench.json'))
print("value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"])
print("roofline", {k: d["roofline"][k] for k in ("frac","frac_survey_5Bpx","launch_ms","alone","traffic")})
print("fcn", {k: d["fcn"].get(k) for k in ("precision","ms_per_frame","algorithmic_tflops","max_abs_logit_diff_vs_oracle")})
print("e2e", d.get("e2e_rgb"))
PY
LM_BENCH_REHEARSE=1 timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --e2e-frames 32 > $O/bench_rehearse_n2.json 2> $O/bench_rehearse_n2.err || { tail -30 $O/bench_rehearse_n2.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3d/bench_rehearse_n2.json').read().strip().splitlines()[-1])
print("N=2 rehearsal value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"] if d["parity"] else None)
print("amdahl", d["amdahl"]); print("per rank", d["per_rank_step_ms"]); print("rgb", d["rgb_sharded"])
PY
