#!/bin/bash
# round 4, job U: slices per weight group of the three large full-resolution layers
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4u; mkdir -p $O
run() { # name env...
  name=$1; shift
  cd /tmp
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$name -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_$name.log 2>&1 || { tail -5 $O/p_$name.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/fcn_layers.py $(find $O/p_$name -name "*_results.db" | head -1) > $O/r04_gsize_$name.txt
  rm -rf $O/p_$name
  echo "== $name ($@)"; grep "3, 3, 4, 2\|7, 7\|frame total" $O/r04_gsize_$name.txt
}
run default LM_X=1
run g2 LM_FCN2_GSIZE=18=2,15=2,19=2
run g3 LM_FCN2_GSIZE=18=3,15=3,19=3
run g7 LM_FCN2_GSIZE=18=7,15=5,19=7
run default2 LM_X=1
