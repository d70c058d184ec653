#!/bin/bash
# round 4, final evidence with the final code: the driver's bench invocation, its kernel stats, FCN per layer + MFMA-pipe counters + formats,
# the 4K line, the two-rank rehearsal started by bench.py itself
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4final; mkdir -p $O; R=$GRAFT_REPO_ROOT
timeout -k 10 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_final_bench_driver_like.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
python3 - $O/r04_final_bench_driver_like.json <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print("value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"])
print("roofline", {k: d["roofline"][k] for k in ("frac","frac_survey_5Bpx","launch_ms","alone","traffic")})
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["sparse_prefix"]["value"])
print("fcn", {k: d["fcn"].get(k) for k in ("precision","ms_per_frame","algorithmic_tflops","frac_of_peak_algorithmic","executed_tflops","max_abs_logit_diff_vs_oracle","binary_flips_vs_oracle")})
print("e2e", {k: d["e2e_rgb"].get(k) for k in ("value","ms_per_frame","fcn_engines_in_flight")})
PY
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/p_bench -o b -- python3 $R/bench.py --gpus 1 --steps 6 --warmup 1 --fcn-frames 0 --cpu-frames 0 > $O/p_bench.log 2>&1
python3 $R/tools/db_to_stats_csv.py $(find $O/p_bench -name "*_results.db" | head -1) $O/r04_final_bench_kernel_stats.csv; rm -rf $O/p_bench
timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_fcn -o f -- python3 $R/tools/fcn_microbench.py mixed 10 > $O/p_fcn.log 2>&1
python3 $R/tools/fcn_layers.py $(find $O/p_fcn -name "*_results.db" | head -1) > $O/r04_final_fcn_layers_mixed.txt; rm -rf $O/p_fcn; tail -1 $O/p_fcn.log; tail -1 $O/r04_final_fcn_layers_mixed.txt
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/m -- python3 $R/bench.py --workload fcn --steps 2 --warmup 1 --no-fcn-oracle > $O/m.log 2>&1 || { tail -5 $O/m.log; exit 1; }
cd $R
python3 tools/fcn_mfma_pmc.py $(ls $O/m/*/*counter_collection.csv | head -1) $O/r04_final_fcn_mfma_pmc_mixed.json | tail -2; rm -rf $O/m
timeout -k 10 600 python tools/fcn_formats.py $O/r04_final_fcn_formats.json 3 30 "only=mixed;r3-mixed (all six f16x3);px2=a2;planar-f16" > $O/r04_final_fcn_formats.txt 2>&1 || { tail -20 $O/r04_final_fcn_formats.txt; exit 1; }
grep -v amdgpu.ids $O/r04_final_fcn_formats.txt | tail -4
timeout -k 10 600 python bench.py --height 2160 --width 3840 --frames 1024 --batch 16 --steps 5 --warmup 4 --fcn-frames 20 --e2e-frames 32 > $O/r04_final_4k_bench.json 2> $O/b4k.err || { tail -20 $O/b4k.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/r04_final_4k_bench.json')); print('4K value', d['value'], 'parity', d['parity']['match'], 'frac', d['roofline']['frac'], 'alone', d['roofline']['alone']['frac'], 'traffic', d['roofline']['traffic'], 'e2e', d['e2e_rgb'].get('value'))"
LM_BENCH_REHEARSE=1 timeout -k 10 400 python bench.py --gpus 2 --frames 1024 --steps 3 --warmup 2 --e2e-frames 8 > $O/r04_final_bench_rehearse_n2.json 2> $O/rehearse.err || { tail -20 $O/rehearse.err; exit 1; }
python3 -c "
import json; d=json.load(open('$O/r04_final_bench_rehearse_n2.json')); print('N=2 rehearsal: value', d['value'], 'parity', d['parity'] and d['parity']['match'], 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'], 'traffic', d['roofline']['traffic'], 'rgb', d['rgb_sharded'] and d['rgb_sharded']['value'])"
