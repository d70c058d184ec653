#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/fcnx; mkdir -p $O
for v in default w0 NOEPI PATCH0; do
  if [ $v = default ]; then unset LM_LIB_PATH; else export LM_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/liblm_$v.so; fi
  cd /tmp
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $O/p_$v -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py f16x3 10 > $O/$v.log 2>&1
  cd $GRAFT_REPO_ROOT
  grep "ms/frame" $O/$v.log
  python3 tools/fcn_layers.py $(find $O/p_$v -name "*_results.db" | head -1) > $O/layers_$v.txt
  rm -rf $O/p_$v
done
paste <(cut -c1-40,62- $O/layers_default.txt) <(cut -c62- $O/layers_w0.txt) <(cut -c62- $O/layers_NOEPI.txt) <(cut -c62- $O/layers_PATCH0.txt)
