#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3m; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_cc_gpu.py tests/test_stream1080p_gpu.py tests/test_dropin_gpu.py -x -q -m gpu -k "not fcn" > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -2 $O/gpu_tests.txt
LM_GROUP_TIMING=1 timeout -k 10 300 python tools/chain_profile.py 2>&1 | grep -E "rep|gimg: max" | tail -6
run() { name=$1; shift
  env "$@" timeout -k 10 400 python bench.py --gpus 1 --steps 9 --warmup 3 --fcn-frames 0 --cpu-frames 0 > $O/bench_$name.json 2> $O/bench_$name.err || { tail -5 $O/bench_$name.err; return 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3m/bench_$name.json')); r=d['roofline']
print('$name', 'value', d['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'parity', d['parity']['match'])"
}
run a X=1 && run b X=1
