#!/bin/bash
# round 4: the whole GPU suite (one process), then the labelling launch alone with 32- and 64-row bands
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4tests; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=15 > $O/r04_gpu_tests.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -25 $O/r04_gpu_tests.txt
[ $rc -eq 0 ] || exit 1
for R in 32 64; do
  echo "== LM_BAND_ROWS=$R"
  LM_BAND_ROWS=$R LM_LABEL_PARTS=1 timeout -k 10 120 python tools/label_microbench.py 64 1080 1920 5000 2>&1 | grep -v amdgpu.ids | tail -3 | tee $O/r04_label_band_rows_$R.txt
done
