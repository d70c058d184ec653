#!/bin/bash
# round 4, job V: 16 x 32 tiles with two channel tiles for the f16 layers below full resolution
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4v; mkdir -p $O
run() { # name env...
  name=$1; shift
  cd /tmp
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$name -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_$name.log 2>&1 || { tail -5 $O/p_$name.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/fcn_layers.py $(find $O/p_$name -name "*_results.db" | head -1) > $O/r04_widedeep_$name.txt
  rm -rf $O/p_$name
  echo "== $name ($@)"; grep -v "copyBuffer\|convT_border\|prepare\|nhwc" $O/r04_widedeep_$name.txt | awk '{printf "%s ", $NF=="us" ? $(NF-1) : $0} END {print ""}'
}
run default LM_X=1
run wide_mt2 LM_FCN_VARIANTS=1=2:0,2=2:0,3=2:0,4=2:0,12=2:0,13=2:0,14=2:0 LM_FCN2_MT=1=2,2=2,3=2,4=2,12=2,13=2
run wide_mt3 LM_FCN_VARIANTS=1=2:0,2=2:0,3=2:0,4=2:0,12=2:0,13=2:0,14=2:0 LM_FCN2_MT=1=3,2=3,3=3,4=3,12=3,13=3
