#!/bin/bash
# round 3, job J: where a step's time goes (depth 1, verbose, step-03 phase timing)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3j; mkdir -p $O
LM_BENCH_VERBOSE=1 LM_GROUP_TIMING=1 timeout -k 10 400 python bench.py --gpus 1 --steps 3 --warmup 2 --depth 1 --fcn-frames 0 --cpu-frames 0 > $O/bench_depth1.json 2> $O/bench_depth1.err
tail -70 $O/bench_depth1.err
LM_BENCH_NO_SPLIT=1 timeout -k 10 400 python bench.py --gpus 1 --steps 4 --warmup 2 --depth 1 --fcn-frames 0 --cpu-frames 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('depth1 nosplit value', d['value'], d['ms_per_step'])"
