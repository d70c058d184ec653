#!/bin/bash
# round 4, job A: operand-format sweep of the planar FCN engine (error, binary flips vs the oracle, ms/frame), engines on 1-3 streams
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4a; mkdir -p $O
timeout -k 10 900 python tools/fcn_formats.py $O/r04_fcn_formats.json 3 30 > $O/r04_fcn_formats.txt 2>&1 || { tail -20 $O/r04_fcn_formats.txt; exit 1; }
cat $O/r04_fcn_formats.txt
timeout -k 10 200 python tools/fcn_two_streams.py 48 2>&1 | tee $O/two_streams.txt
