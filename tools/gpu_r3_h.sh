#!/bin/bash
# round 3, job H: staggered label parts: microbench + bench
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3h; mkdir -p $O
for S in 1 0; do echo "LM_LABEL_STAGGER=$S"; LM_LABEL_STAGGER=$S timeout -k 10 200 python tools/label_microbench.py 64 1080 1920 5000 2>&1 | grep -E "us/launch"; done | tee $O/label_stagger.txt
LM_LABEL_PARTS=1 timeout -k 10 200 python tools/label_microbench.py 64 1080 1920 5000 2>&1 | grep -E "FUSED" | tee -a $O/label_stagger.txt
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --gpus 1 --steps 9 --warmup 3 --fcn-frames 0 --cpu-frames 0 > $O/bench_$name.json 2> $O/bench_$name.err || { tail -5 $O/bench_$name.err; return 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3h/bench_$name.json')); r=d['roofline']
print('$name', 'value', d['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'launch_ms', r['launch_ms'], 'alone', r.get('alone',{}).get('frac'), 'parity', d['parity']['match'])"
}
run warm LM_LABEL_STAGGER=1 && run stagger1 LM_LABEL_STAGGER=1 && run stagger0 LM_LABEL_STAGGER=0 && run stagger1b LM_LABEL_STAGGER=1 && run stagger1_free LM_LABEL_STAGGER=1 LM_BENCH_SCHEDULE=free
