#!/bin/bash
# round 4, job K: channel tiles per workgroup of the thin / deep layers
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4k; mkdir -p $O
run() { # name env...
  name=$1; shift
  cd /tmp
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$name -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_$name.log 2>&1 || { tail -5 $O/p_$name.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/fcn_layers.py $(find $O/p_$name -name "*_results.db" | head -1) > $O/r04_mt_$name.txt
  rm -rf $O/p_$name
  echo "== $name ($@)"; tail -1 $O/p_$name.log; grep -v "copyBuffer\|convT_border\|7, 7\|1, 7\|vsum\|prepare\|nhwc" $O/r04_mt_$name.txt
}
run b LM_FCN2_MT=0=1,1=1,2=3,3=3,4=3,12=3,13=2,5=4,11=4
