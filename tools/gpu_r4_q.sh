#!/bin/bash
# round 4, job Q: start-up stagger of the CUs' second resident workgroups
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4q; mkdir -p $O
run() { # name env...
  name=$1; shift
  cd /tmp
  env "$@" timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 30 2>&1 | tail -1
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$name -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_$name.log 2>&1 || { tail -5 $O/p_$name.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/fcn_layers.py $(find $O/p_$name -name "*_results.db" | head -1) > $O/r04_stagger_$name.txt
  rm -rf $O/p_$name
  echo "== $name ($@)"; grep -v "copyBuffer\|convT_border\|prepare\|nhwc" $O/r04_stagger_$name.txt | awk '{printf "%s ", $NF=="us" ? $(NF-1) : $0} END {print ""}'
}
run s0 LM_G2_STAGGER=0
run s2 LM_G2_STAGGER=2
run s4 LM_G2_STAGGER=4
run s6 LM_G2_STAGGER=6
run s0b LM_G2_STAGGER=0
