#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/nt; mkdir -p $O
for v in 384 100000000 1 384; do
  export LM_FCN_NT2_MIN_BLOCKS=$v
  cd /tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/p_$v -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py f16x3 20 > $O/$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  echo "NT2_MIN=$v $(grep 'ms/frame' $O/$v.log)"
  python3 tools/fcn_layers.py $(find $O/p_$v -name "*_results.db" | head -1) > $O/layers_$v.txt
  rm -rf $O/p_$v
done
paste <(cut -c1-40,62- $O/layers_384.txt) <(cut -c62- $O/layers_100000000.txt) <(cut -c62- $O/layers_1.txt) | head -20
