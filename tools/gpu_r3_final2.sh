#!/bin/bash
# round 3, session 2, final evidence: GPU suite, smoke, the driver's bench invocation (twice: the first process on a fresh lease runs
# slower), kernel stats + timeline of the bench, labelling launch alone with per-kernel averages
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3final2; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/r03_final_gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -1 $O/r03_final_gpu_tests.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for rep in 1 2; do
timeout -k 10 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_final_bench_driver_like_$rep.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - $O/r03_final_bench_driver_like_$rep.json <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print("value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"])
print("roofline", {k: d["roofline"][k] for k in ("frac","frac_survey_5Bpx","launch_ms","alone","traffic")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["dense_window"]["value"])
print("fcn", {k: d["fcn"].get(k) for k in ("precision","ms_per_frame","algorithmic_tflops","max_abs_logit_diff_vs_oracle")}, "e2e", d["e2e_rgb"]["value"])
PY
done
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/p_bench -o b -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 6 --warmup 1 --fcn-frames 0 --cpu-frames 0 > $O/p_bench.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p_label -o l -- python3 $GRAFT_REPO_ROOT/tools/label_microbench.py 64 1080 1920 5000 > $O/r03_final_label_microbench.txt 2>&1
cd $GRAFT_REPO_ROOT
DB=$(find $O/p_bench -name "*_results.db" | head -1)
python3 tools/db_to_stats_csv.py $DB $O/r03_final_bench_kernel_stats.csv
python3 tools/prof_timeline.py $DB > $O/r03_final_bench_timeline.txt; cat $O/r03_final_bench_timeline.txt
python3 tools/db_to_stats_csv.py $(find $O/p_label -name "*_results.db" | head -1) $O/r03_final_label_microbench_kernel_stats.csv
grep -v "^W2\|^E2" $O/r03_final_label_microbench.txt | tail -4
rm -rf $O/p_bench $O/p_label
