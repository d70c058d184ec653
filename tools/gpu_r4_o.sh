#!/bin/bash
# round 4, job O: loader wave for the three- and four-tile layers now that the lean epilogue leaves them at <= 152 VGPRs
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4o; mkdir -p $O
run() { # name env...
  name=$1; shift
  cd /tmp
  env "$@" timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 30 2>&1 | tail -1
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$name -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_$name.log 2>&1 || { tail -5 $O/p_$name.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/fcn_layers.py $(find $O/p_$name -name "*_results.db" | head -1) > $O/r04_loader2_$name.txt
  rm -rf $O/p_$name
  echo "== $name ($@)"; grep -v "copyBuffer\|convT_border\|prepare\|nhwc" $O/r04_loader2_$name.txt | awk '{printf "%s ", $NF=="us" ? $(NF-1) : $0} END {print ""}'
}
run now LM_X=1
run loader_all LM_FCN_VARIANTS=1=1:1,2=1:1,3=1:1,4=1:1,12=1:1,13=1:1,14=1:1
run now2 LM_X=1
run loader_all2 LM_FCN_VARIANTS=1=1:1,2=1:1,3=1:1,4=1:1,12=1:1,13=1:1,14=1:1
