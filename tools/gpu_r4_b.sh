#!/bin/bash
# round 4, job B: where lm_k_g2's workgroups spend their time (diagnostic stamps build), per layer
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4b; mkdir -p $O
for L in 18 19 15 0 5 2 4 11 13 10 16; do
  LM_G2_STAMP_LAYER=$L timeout -k 10 120 python tools/fcn_stamps.py "" 2>&1 | grep -v amdgpu.ids >> $O/r04_fcn_stamps_mixed.txt || exit 1
done
for L in 18 19 15; do
  LM_G2_STAMP_LAYER=$L timeout -k 10 120 python tools/fcn_stamps.py "15=w2,18=w2,19=w2" 2>&1 | grep -v amdgpu.ids >> $O/r04_fcn_stamps_w2.txt || exit 1
  LM_G2_STAMP_LAYER=$L timeout -k 10 120 python tools/fcn_stamps.py "15=a2,18=a2,19=a2" 2>&1 | grep -v amdgpu.ids >> $O/r04_fcn_stamps_a2.txt || exit 1
done
cat $O/r04_fcn_stamps_mixed.txt $O/r04_fcn_stamps_w2.txt $O/r04_fcn_stamps_a2.txt
