#!/bin/bash
# round 4, job P: stamps after the lean epilogue / loop-tail changes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4p; mkdir -p $O
for L in 18 19 15 1 3 10; do
  LM_G2_STAMP_LAYER=$L timeout -k 10 120 python tools/fcn_stamps.py "" 2>&1 | grep -v amdgpu.ids >> $O/r04_fcn_stamps_v5_lean_epilogue.txt || exit 1
done
cat $O/r04_fcn_stamps_v5_lean_epilogue.txt
