#!/bin/bash
# round 3, job B: the planar FCN engine on the GPU: parity vs the oracle, time per frame, per-layer times
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3b; mkdir -p $O
timeout -k 10 300 python tools/fcn_planar_check.py > $O/planar_check.txt 2>&1 || { tail -20 $O/planar_check.txt; exit 1; }
cat $O/planar_check.txt
for P in mixed planar-f16x3 planar-f16 f16x3; do timeout -k 10 120 python tools/fcn_microbench.py $P 20 2>&1 | tail -1; done
for P in mixed planar-f16; do
  cd /tmp
  timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$P -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py $P 10 > $O/p_$P.log 2>&1 || { tail -5 $O/p_$P.log; exit 1; }
  cd $GRAFT_REPO_ROOT
  python3 tools/fcn_layers.py $(find $O/p_$P -name "*_results.db" | head -1) > $O/fcn_layers_$P.txt
  rm -rf $O/p_$P
  cat $O/fcn_layers_$P.txt
done
