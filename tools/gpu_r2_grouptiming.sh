#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/gt; mkdir -p $O
LM_GROUP_TIMING=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --depth 1 --fcn-frames 0 --cpu-frames 0 > $O/d1.json 2> $O/d1.err; grep -i -E "group|ms" $O/d1.err | tail -40
