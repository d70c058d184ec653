#!/bin/bash
# round-2 final evidence: full gpu suite, default bench (driver invocation), profiles
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/final; mkdir -p $O
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > $O/gpu_tests.txt; cat $O/gpu_tests.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r02_bench_driver_like.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --gpus 1 --steps 6 --warmup 2 --depth 1 --fcn-frames 0 --cpu-frames 0 > $O/r02_bench_depth1.json 2>/dev/null
LM_BENCH_SCHEDULE=gated python bench.py --gpus 1 --steps 6 --warmup 2 --fcn-frames 0 --cpu-frames 0 > $O/r02_bench_gated.json 2>/dev/null
LM_BENCH_REHEARSE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --cpu-frames 0 --fcn-frames 0 > $O/r02_bench_rehearse_n2.json 2> $O/rehearse.err; echo "rehearse rc=$?"
python tools/label_microbench.py 64 1080 1920 5000 > $O/r02_label_microbench.txt 2>&1
python tools/label_microbench.py 64 1080 1920 192 >> $O/r02_label_microbench.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/p_bench -o b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --fcn-frames 0 --cpu-frames 0 > $O/p_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/p_label -o l -- python3 $GRAFT_REPO_ROOT/tools/label_microbench.py 64 1080 1920 5000 > $O/p_label.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/p_fcn -o f -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcn --steps 10 --no-fcn-oracle > $O/p_fcn.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/db_to_stats_csv.py $(find $O/p_bench -name "*_results.db" | head -1) $O/r02_final_bench_kernel_stats.csv
python3 tools/db_to_stats_csv.py $(find $O/p_label -name "*_results.db" | head -1) $O/r02_final_label_microbench_kernel_stats.csv
python3 tools/db_to_stats_csv.py $(find $O/p_fcn -name "*_results.db" | head -1) $O/r02_final_fcn_f16x3_kernel_stats.csv
rm -rf $O/p_bench $O/p_label $O/p_fcn
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/final/r02_bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], "roof", d["roofline"]["frac"], d["roofline"].get("alone",{}).get("frac"), "parity", d["parity"]["match"], "fcn", d.get("fcn",{}).get("ms_per_frame"), "e2e", d.get("e2e_rgb",{}).get("value"))
    except Exception as e: print(f, "ERR", e)
PY
cat $O/r02_label_microbench.txt | grep labels=True
