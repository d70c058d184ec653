#!/bin/bash
# round 3, job K: lm_k_middle (seam + flatten + numbering as one launch with per-frame rendezvous) against the three launches
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3k; mkdir -p $O
LM_LABEL_FUSED_MIDDLE=1 timeout -k 10 600 python -m pytest tests/test_cc_gpu.py tests/test_stream1080p_gpu.py -x -q -m gpu -k "not fcn" > $O/gpu_tests_fused_middle.txt 2>&1; echo "pytest (fused middle) rc=$?"; tail -2 $O/gpu_tests_fused_middle.txt
for M in 0 1; do for P in 1 2; do echo "LM_LABEL_FUSED_MIDDLE=$M LM_LABEL_PARTS=$P"; LM_LABEL_FUSED_MIDDLE=$M LM_LABEL_PARTS=$P timeout -k 10 200 python tools/label_microbench.py 64 1080 1920 5000 2>&1 | grep -E "us/launch"; done; done | tee $O/label_fused_middle.txt
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --gpus 1 --steps 9 --warmup 3 --fcn-frames 0 --cpu-frames 0 > $O/bench_$name.json 2> $O/bench_$name.err || { tail -5 $O/bench_$name.err; return 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3k/bench_$name.json')); r=d['roofline']
print('$name', 'value', d['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'launch_ms', r['launch_ms'], 'alone', r.get('alone',{}).get('frac'), 'parity', d['parity']['match'])"
}
run warm LM_LABEL_FUSED_MIDDLE=0 && run three LM_LABEL_FUSED_MIDDLE=0 && run fused LM_LABEL_FUSED_MIDDLE=1 && run three_b LM_LABEL_FUSED_MIDDLE=0 && run fused_b LM_LABEL_FUSED_MIDDLE=1
cd /tmp
LM_LABEL_FUSED_MIDDLE=1 LM_LABEL_PARTS=1 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/p_label -o l -- python3 $GRAFT_REPO_ROOT/tools/label_microbench.py 64 1080 1920 5000 > $O/p_label.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/db_to_stats_csv.py $(find $O/p_label -name "*_results.db" | head -1) $O/label_microbench_fused_middle_kernel_stats.csv
rm -rf $O/p_label
head -9 $O/label_microbench_fused_middle_kernel_stats.csv | cut -c1-40,150-
