#!/bin/bash
# round 4, job D: stamps v2 (DMA issue / slice loops / vmcnt / barrier) on the deep layers and the full-resolution ones; ring 2 vs 3
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4d; mkdir -p $O
for L in 5 11 4 2 13 1 14 18 19 15; do
  LM_G2_STAMP_LAYER=$L timeout -k 10 120 python tools/fcn_stamps.py "15=w2,18=w2,16=w2" 2>&1 | grep -v amdgpu.ids >> $O/r04_fcn_stamps_v2.txt || exit 1
done
for L in 5 11; do
  LM_FCN2_RING=2 LM_G2_STAMP_LAYER=$L timeout -k 10 120 python tools/fcn_stamps.py "15=w2,18=w2,16=w2" 2>&1 | grep -v amdgpu.ids >> $O/r04_fcn_stamps_v2_ring2.txt || exit 1
done
cat $O/r04_fcn_stamps_v2.txt; echo RING2; cat $O/r04_fcn_stamps_v2_ring2.txt
