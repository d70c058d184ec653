#!/bin/bash
# round 4, job C: the slice loop with counted LDS waits (buffer_load ... lds + LDS slice table): FCN parity tests, format sweep, per-layer times
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4c; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "fcn" > $O/gpu_tests_fcn.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/gpu_tests_fcn.txt
timeout -k 10 600 python tools/fcn_formats.py $O/r04_fcn_formats_v2.json 3 30 "only=mixed;up1,px1=w2;up1,px1=a2;up1,px1,txt=w2;up1,px1,px2,txt=w2;planar-f16" > $O/r04_fcn_formats_v2.txt 2>&1 || { tail -20 $O/r04_fcn_formats_v2.txt; exit 1; }
grep -v amdgpu.ids $O/r04_fcn_formats_v2.txt
cd /tmp
for F in "" "15=w2,18=w2,16=w2"; do
  N=$(echo "mixed_$F" | tr ',=' '__')
  LM_FCN_FORMATS=$F timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$N -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_$N.log 2>&1 || { tail -5 $O/p_$N.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/fcn_layers.py $(find $O/p_$N -name "*_results.db" | head -1) > $O/r04_v2_fcn_layers_$N.txt
  rm -rf $O/p_$N
  tail -1 $O/p_$N.log; cat $O/r04_v2_fcn_layers_$N.txt
done
