#!/bin/bash
# round 3, job G: pipeline depth of the N = 1 bench
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3g; mkdir -p $O
run() { name=$1; shift
  timeout -k 10 400 python bench.py --gpus 1 --steps 9 --warmup 3 --fcn-frames 0 --cpu-frames 0 "$@" > $O/bench_$name.json 2> $O/bench_$name.err || { tail -5 $O/bench_$name.err; return 1; }
  python3 -c "
import json; d=json.load(open('gpurun_out/r3g/bench_$name.json')); r=d['roofline']
print('$name', 'value', d['value'], 'ms/step', d['ms_per_step'], 'frac', r['frac'], 'launch_ms', r['launch_ms'], 'parity', d['parity']['match'])"
}
run warm --depth 2 && run depth2 --depth 2 && run depth3 --depth 3 && run depth4 --depth 4 && run depth1 --depth 1 && run depth3b --depth 3
