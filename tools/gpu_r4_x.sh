#!/bin/bash
# round 4, job X: conv_pixels_1 / _2 with one-octet chunks and double-buffered patch planes
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4x; mkdir -p $O
run() { # name env...
  name=$1; shift
  cd /tmp
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$name -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_$name.log 2>&1 || { tail -5 $O/p_$name.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/fcn_layers.py $(find $O/p_$name -name "*_results.db" | head -1) > $O/r04_pxchunks_$name.txt
  rm -rf $O/p_$name
  echo "== $name ($@)"; grep "7, 7\|frame total" $O/r04_pxchunks_$name.txt
}
run default LM_X=1
run px1_oct1 LM_FCN2_PX1_OCTETS=1
run px1_oct1_pd LM_FCN2_PX1_OCTETS=1 LM_FCN2_PX_PDOUBLE=1,0
run px1_pd_lds LM_FCN2_PX1_OCTETS=1 LM_FCN2_PX_PDOUBLE=1,0 LM_FCN2_LDS=18=81920
run default2 LM_X=1
