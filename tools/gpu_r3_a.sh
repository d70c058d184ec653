#!/bin/bash
# round 3, job A: GPU parity suite on the round-3 fixes, per-layer precision attribution, per-layer times of the three formats
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3a; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "pytest rc=$?"; tail -3 $O/gpu_tests.txt
timeout -k 10 400 python tools/fcn_layer_precision.py $O/fcn_layer_precision.json 3 > $O/fcn_layer_precision.txt 2>&1 || { tail -20 $O/fcn_layer_precision.txt; exit 1; }
tail -50 $O/fcn_layer_precision.txt
for P in f16x3 f16x2 f16; do
  cd /tmp
  timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$P -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py $P 10 > $O/p_$P.log 2>&1 || { tail -5 $O/p_$P.log; exit 1; }
  cd $GRAFT_REPO_ROOT
  python3 tools/fcn_layers.py $(find $O/p_$P -name "*_results.db" | head -1) > $O/fcn_layers_$P.txt
  rm -rf $O/p_$P
  tail -1 $O/p_$P.log; tail -1 $O/fcn_layers_$P.txt
done
