#!/bin/bash
# round 4, job H: bench as the driver runs it, two-rank rehearsal started by bench.py itself, stamps of the full-resolution layers
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4h; mkdir -p $O
timeout -k 10 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_mid_bench_driver_like.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - $O/r04_mid_bench_driver_like.json <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print("value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"])
print("roofline", {k: d["roofline"][k] for k in ("frac","frac_survey_5Bpx","launch_ms","alone","traffic")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sparse_prefix"]["value"])
print("fcn", {k: d["fcn"].get(k) for k in ("precision","ms_per_frame","algorithmic_tflops","frac_of_peak_algorithmic","max_abs_logit_diff_vs_oracle","binary_flips_vs_oracle")})
print("e2e", d["e2e_rgb"])
PY
LM_BENCH_REHEARSE=1 timeout -k 10 400 python bench.py --gpus 2 --frames 1024 --steps 3 --warmup 2 --e2e-frames 8 > $O/r04_bench_rehearse_n2.json 2> $O/rehearse.err || { tail -20 $O/rehearse.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/r04_bench_rehearse_n2.json')); print('N=2 rehearsal: value', d['value'], 'parity', d['parity'] and d['parity']['match'], 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'], 'traffic', d['roofline']['traffic'], 'rgb', d['rgb_sharded'] and d['rgb_sharded']['value'])"
for L in 18 19 15 0 10 16; do
  LM_G2_STAMP_LAYER=$L timeout -k 10 120 python tools/fcn_stamps.py "" 2>&1 | grep -v amdgpu.ids >> $O/r04_fcn_stamps_v4_defaults.txt || exit 1
done
cat $O/r04_fcn_stamps_v4_defaults.txt
