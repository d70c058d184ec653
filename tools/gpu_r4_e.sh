#!/bin/bash
# round 4, job E: loader-wave kernel: FCN parity tests, timing of the candidate assignments, per-layer times, stamps
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4e; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "fcn" > $O/gpu_tests_fcn.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/gpu_tests_fcn.txt
timeout -k 10 600 python tools/fcn_formats.py $O/r04_fcn_formats_v3.json 1 30 "only=mixed;up1,px1=w2;up1,px1,txt=w2;planar-f16" > $O/r04_fcn_formats_v3.txt 2>&1 || { tail -20 $O/r04_fcn_formats_v3.txt; exit 1; }
grep -v amdgpu.ids $O/r04_fcn_formats_v3.txt
cd /tmp
for F in "15=w2,18=w2,16=w2"; do
  N=$(echo "mixed_$F" | tr ',=' '__')
  LM_FCN_FORMATS=$F timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$N -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_$N.log 2>&1 || { tail -5 $O/p_$N.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/fcn_layers.py $(find $O/p_$N -name "*_results.db" | head -1) > $O/r04_v3_fcn_layers_$N.txt
  rm -rf $O/p_$N
  tail -1 $O/p_$N.log; cat $O/r04_v3_fcn_layers_$N.txt
done
cd $GRAFT_REPO_ROOT
for L in 5 11 4 18 19 15; do
  LM_G2_STAMP_LAYER=$L timeout -k 10 120 python tools/fcn_stamps.py "15=w2,18=w2,16=w2" 2>&1 | grep -v amdgpu.ids >> $O/r04_fcn_stamps_v3.txt || exit 1
done
cat $O/r04_fcn_stamps_v3.txt
