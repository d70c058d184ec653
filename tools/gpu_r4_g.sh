#!/bin/bash
# round 4, job G: final format sweep (3 seeds), px2 16x32 tiles with one-octet chunks, 150 KB LDS for the one-workgroup-per-CU layers, streams
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4g; mkdir -p $O
run() { # name env...
  name=$1; shift
  cd /tmp
  env "$@" timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 30 2>&1 | tail -1
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$name -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_$name.log 2>&1 || { tail -5 $O/p_$name.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/fcn_layers.py $(find $O/p_$name -name "*_results.db" | head -1) > $O/r04_variants2_$name.txt
  rm -rf $O/p_$name
  echo "== $name ($@)"; grep -v "copyBuffer\|convT_border" $O/r04_variants2_$name.txt
}
run default LM_X=1
run px2wide LM_FCN2_PX_OCTETS=1 LM_FCN_VARIANTS=19=2:0
run px2oct1 LM_FCN2_PX_OCTETS=1
run deeplds LM_FCN2_DEEP_LDS=153600
run deeplds_loader LM_FCN2_DEEP_LDS=153600 LM_FCN_VARIANTS=4=1:1,12=1:1
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python tools/fcn_formats.py $O/r04_fcn_formats.json 3 30 > $O/r04_fcn_formats.txt 2>&1 || { tail -20 $O/r04_fcn_formats.txt; exit 1; }
grep -v amdgpu.ids $O/r04_fcn_formats.txt
timeout -k 10 200 python tools/fcn_two_streams.py 48 2>&1 | grep -v amdgpu.ids | tee $O/two_streams.txt
