#!/bin/bash
# round 3, final: the GPU suite, the driver's bench invocation, bench kernel stats, FCN per-layer times + counters
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3final; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/r03_final_gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/r03_final_gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_final_bench_driver_like.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3final/r03_final_bench_driver_like.json'))
print("value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"])
print("roofline", {k: d["roofline"][k] for k in ("frac","frac_survey_5Bpx","launch_ms","alone","traffic")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["dense_window"]["value"])
print("fcn", {k: d["fcn"].get(k) for k in ("precision","ms_per_frame","algorithmic_tflops","executed_tflops","max_abs_logit_diff_vs_oracle")})
print("e2e", d["e2e_rgb"]["value"])
PY
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/p_bench -o b -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 5 --warmup 1 --fcn-frames 0 --cpu-frames 0 > $O/p_bench.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_fcn -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_fcn.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/db_to_stats_csv.py $(find $O/p_bench -name "*_results.db" | head -1) $O/r03_final_bench_kernel_stats.csv
python3 tools/fcn_layers.py $(find $O/p_fcn -name "*_results.db" | head -1) > $O/r03_final_fcn_layers_mixed.txt
rm -rf $O/p_bench $O/p_fcn
tail -1 $O/p_fcn.log; tail -1 $O/r03_final_fcn_layers_mixed.txt
