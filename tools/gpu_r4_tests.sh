#!/bin/bash
# round 4: the whole GPU suite (one process) + smoke
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4tests; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=8 > $O/r04_final_gpu_tests.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -18 $O/r04_final_gpu_tests.txt
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
