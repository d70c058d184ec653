#!/bin/bash
# round 4, job M: fused heads (row convolution + vertical sum in one kernel) and the lean epilogue: FCN tests, A/B per layer
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4m; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "fcn or e2e" > $O/gpu_tests_fcn.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/gpu_tests_fcn.txt
[ $rc -eq 0 ] || { grep -n "Error\|assert" $O/gpu_tests_fcn.txt | head; exit 1; }
run() { # name env...
  name=$1; shift
  cd /tmp
  env "$@" timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 30 2>&1 | tail -1
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$name -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_$name.log 2>&1 || { tail -5 $O/p_$name.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/fcn_layers.py $(find $O/p_$name -name "*_results.db" | head -1) > $O/r04_heads_$name.txt
  rm -rf $O/p_$name
  echo "== $name ($@)"; grep "1, 7\|vsum\|frame total\|nhwc\|copyBuffer" $O/r04_heads_$name.txt
}
run unfused LM_FCN2_FUSED_HEADS=0
run fused LM_X=1
run unfused2 LM_FCN2_FUSED_HEADS=0
run fused2 LM_X=1
