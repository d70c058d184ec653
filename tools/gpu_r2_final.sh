#!/bin/bash
# round-2 final evidence: full gpu suite, smoke, default bench (driver invocation), schedules, FCN ladder, profiles, counters
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/final2; mkdir -p $O
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
grep -q passed $O/gpu_tests.txt || exit 1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r02_bench_driver_like.json 2> $O/bench.err; echo "bench rc=$?"
LM_BENCH_SCHEDULE=free python bench.py --gpus 1 --steps 6 --warmup 2 --fcn-frames 0 --cpu-frames 0 > $O/r02_bench_free.json 2>/dev/null
python bench.py --gpus 1 --steps 6 --warmup 2 --depth 1 --fcn-frames 0 --cpu-frames 0 > $O/r02_bench_depth1.json 2>/dev/null
LM_BENCH_REHEARSE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-frames 0 --fcn-frames 0 > $O/r02_bench_rehearse_n2.json 2> $O/rehearse.err; echo "rehearse rc=$?"
for p in f16x3 f16x2 f16 fp32; do python bench.py --workload fcn --steps 20 --fcn-precision $p > $O/r02_fcn_precision_ladder_$p.json 2>/dev/null; done
python tools/label_microbench.py 64 1080 1920 5000 > $O/r02_label_microbench.txt 2>&1
python tools/label_microbench.py 64 1080 1920 192 >> $O/r02_label_microbench.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/p_bench -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --fcn-frames 0 --cpu-frames 0 > $O/p_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/p_label -o l -- python3 $GRAFT_REPO_ROOT/tools/label_microbench.py 64 1080 1920 5000 > $O/p_label.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/p_fcn -o f -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcn --steps 10 --no-fcn-oracle > $O/p_fcn.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcn --steps 2 --warmup 1 --no-fcn-oracle > $O/mfma.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/stall -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcn --steps 1 --warmup 1 --no-fcn-oracle > $O/stall.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $GRAFT_REPO_ROOT/tools/label_microbench.py 64 1080 1920 5000 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $GRAFT_REPO_ROOT/tools/label_microbench.py 64 1080 1920 5000 > $O/write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/db_to_stats_csv.py $(find $O/p_bench -name "*_results.db" | head -1) $O/r02_final_bench_kernel_stats.csv
python3 tools/db_to_stats_csv.py $(find $O/p_label -name "*_results.db" | head -1) $O/r02_final_label_microbench_kernel_stats.csv
python3 tools/db_to_stats_csv.py $(find $O/p_fcn -name "*_results.db" | head -1) $O/r02_final_fcn_f16x3_kernel_stats.csv
python3 tools/fcn_layers.py $(find $O/p_fcn -name "*_results.db" | head -1) > $O/r02_final_fcn_f16x3_layers.txt
python3 tools/fcn_mfma_pmc.py $(ls $O/mfma/*/*counter_collection.csv | head -1) $O/r02_fcn_mfma_pmc_f16x3.json
python3 tools/fcn_stall_pmc.py $(ls $O/stall/*/*counter_collection.csv | head -1) $O/r02_fcn_stalls_pmc.txt > /dev/null
python3 tools/pmc_traffic.py $(ls $O/fetch/*/*counter_collection.csv | head -1) $(ls $O/write/*/*counter_collection.csv | head -1) $O/r02_label_traffic_pmc.json 64
rm -rf $O/p_bench $O/p_label $O/p_fcn $O/mfma $O/stall $O/fetch $O/write
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/final2/r02_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], "roof", d["roofline"]["frac"], d["roofline"].get("alone",{}).get("frac"), "parity", d["parity"]["match"], "fcn", d.get("fcn",{}).get("ms_per_frame"), "e2e", d.get("e2e_rgb",{}).get("value"))
    except Exception as e: print(f, "ERR", e)
for f in sorted(glob.glob("gpurun_out/final2/r02_fcn_precision_ladder_*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1])["fcn"]; print(f.split("_")[-1], d["ms_per_frame"], d.get("max_abs_logit_diff_vs_oracle"))
PY
grep labels=True $O/r02_label_microbench.txt
