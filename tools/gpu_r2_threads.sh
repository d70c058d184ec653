#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/thr; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_stream1080p_gpu.py tests/test_cc_gpu.py -x -q -m gpu -k "stream or grouping or 4k" 2>&1 | tail -3 || exit 1
LM_GROUP_TIMING=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --depth 1 --fcn-frames 0 --cpu-frames 0 > $O/d1.json 2> $O/d1.err; grep "lm_group\]" $O/d1.err | tail -22
