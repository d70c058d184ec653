#!/bin/bash
# round 4, job N: slice-loop tail without the accumulator copies; output head fused on 16 x 32 tiles
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4n; mkdir -p $O
run() { # name env...
  name=$1; shift
  cd /tmp
  env "$@" timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 30 2>&1 | tail -1
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$name -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_$name.log 2>&1 || { tail -5 $O/p_$name.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/fcn_layers.py $(find $O/p_$name -name "*_results.db" | head -1) > $O/r04_tail_$name.txt
  rm -rf $O/p_$name
  echo "== $name ($@)"; grep -v "copyBuffer\|convT_border\|prepare\|nhwc" $O/r04_tail_$name.txt | awk '{printf "%s ", $NF=="us" ? $(NF-1) : $0} END {print ""}'
}
run before LM_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/liblm_prev.so
run now LM_X=1
run out_fused_wide LM_FCN2_FUSED_HEADS=3 LM_FCN_VARIANTS=20=2:0
run before2 LM_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/liblm_prev.so
run now2 LM_X=1
