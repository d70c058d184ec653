#!/bin/bash
# round 4, job I: occupancy of the full-resolution layers: 16x16 tiles with 3 / 4 workgroups per CU vs 16x32 tiles with 2
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4i; mkdir -p $O
run() { # name env...
  name=$1; shift
  cd /tmp
  env "$@" timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 30 2>&1 | tail -1
  env "$@" timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$name -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_$name.log 2>&1 || { tail -5 $O/p_$name.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/fcn_layers.py $(find $O/p_$name -name "*_results.db" | head -1) > $O/r04_occupancy_$name.txt
  rm -rf $O/p_$name
  echo "== $name ($@)"; grep "3, 3, 4, 2\|7, 7\|1, 7\|3, 3, 3, 3\|frame total" $O/r04_occupancy_$name.txt
}
run default LM_X=1
run nc1_3wg LM_FCN_VARIANTS=18=1:0,15=1:0,19=1:0 LM_FCN2_LDS=18=53000,15=53000,19=53000 LM_FCN2_PX_OCTETS=1
run nc1_4wg LM_FCN_VARIANTS=18=1:0,15=1:0,19=1:0 LM_FCN2_LDS=18=40000,15=40000,19=40000 LM_FCN2_PX_OCTETS=1
run nc1_3wg_loader LM_FCN_VARIANTS=18=1:1,15=1:1,19=1:1 LM_FCN2_LDS=18=53000,15=53000,19=53000 LM_FCN2_PX_OCTETS=1
run heads_3wg LM_FCN2_LDS=16=53000,20=53000,0=26000
