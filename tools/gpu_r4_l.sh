#!/bin/bash
# round 4, job L: timing-only builds of lm_k_g2 (make -C lecturemath_amd/csrc variants; wrong results, same schedule): what the weight / patch fetches, the activation, the stores and the whole epilogue cost
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4l; mkdir -p $O
for v in default cut1 cut2 cut3 cut4 cut6 cut7; do
  cd /tmp
  if [ $v = default ]; then unset LM_LIB_PATH; else export LM_LIB_PATH=$GRAFT_REPO_ROOT/tools/variants/liblm_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace -d $O/p_$v -o f -- python3 $GRAFT_REPO_ROOT/tools/fcn_microbench.py mixed 10 > $O/p_$v.log 2>&1 || { tail -5 $O/p_$v.log; exit 1; }
  python3 $GRAFT_REPO_ROOT/tools/fcn_layers.py $(find $O/p_$v -name "*_results.db" | head -1) > $O/r04_cuts_$v.txt
  rm -rf $O/p_$v
  echo "== $v"; tail -1 $O/p_$v.log; grep -v "copyBuffer\|convT_border\|vsum\|prepare\|nhwc" $O/r04_cuts_$v.txt | awk '{printf "%s ", $NF=="us" ? $(NF-1) : $0} END {print ""}'
done
