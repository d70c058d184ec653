#!/bin/bash
# round 4, job F: kernel variants (16 x 32 tiles, loader wave) per layer: per-layer times of one forward pass under rocprofv3 --kernel-trace
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4f; mkdir -p $O
export LM_FCN_FORMATS="15=w2,18=w2,16=w2"
run() { # name variants
  cd /tmp
  LM_FCN_VARIANTS="$2" timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$1 -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_$1.log 2>&1 || { tail -5 $O/p_$1.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/fcn_layers.py $(find $O/p_$1 -name "*_results.db" | head -1) > $O/r04_variants_$1.txt
  rm -rf $O/p_$1
  echo "== $1 ($2)"; tail -1 $O/p_$1.log; cat $O/r04_variants_$1.txt
}
run base ""
run wide "0=2:0,1=2:0,14=2:0,15=2:0,16=2:0,18=2:0,19=2:0,20=2:0"
run loader "5=1:1,11=1:1,15=1:1,18=1:1,19=1:1"
run wide_loader "0=2:0,1=2:0,14=2:0,15=2:0,16=2:0,18=2:0,19=2:0,20=2:0,5=1:1,11=1:1"
