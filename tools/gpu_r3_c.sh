#!/bin/bash
# round 3, job C: GPU parity suite, bench (driver invocation), labelling microbench + kernel stats + PMC traffic
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3c; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/gpu_tests.txt
timeout -k 10 600 python bench.py --gpus 1 --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r3c/bench.json'))
print("value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"])
print("roofline", {k: d["roofline"][k] for k in ("frac","frac_survey_5Bpx","launch_ms","alone")})
print("cpu", d["cpu_baseline"])
print("fcn", {k: d["fcn"].get(k) for k in ("precision","ms_per_frame","algorithmic_tflops","max_abs_logit_diff_vs_oracle")})
print("e2e", d.get("e2e_rgb"))
PY
export LM_LABEL_PARTS=1
timeout -k 10 200 python tools/label_microbench.py 64 1080 1920 5000 > $O/label_microbench_parts1.txt 2>&1 || { tail -5 $O/label_microbench_parts1.txt; exit 1; }
cat $O/label_microbench_parts1.txt
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/p_label -o l -- python3 $GRAFT_REPO_ROOT/tools/label_microbench.py 64 1080 1920 5000 > $O/p_label.log 2>&1
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $GRAFT_REPO_ROOT/tools/label_microbench.py 64 1080 1920 5000 > $O/fetch.log 2>&1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $GRAFT_REPO_ROOT/tools/label_microbench.py 64 1080 1920 5000 > $O/write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/db_to_stats_csv.py $(find $O/p_label -name "*_results.db" | head -1) $O/label_microbench_kernel_stats.csv
python3 tools/pmc_traffic.py $(ls $O/fetch/*/*counter_collection.csv | head -1) $(ls $O/write/*/*counter_collection.csv | head -1) $O/r03_label_traffic_pmc.json 64
rm -rf $O/p_label $O/fetch $O/write
head -12 $O/label_microbench_kernel_stats.csv | cut -c1-150
