#!/bin/bash
# round 4, job T: packing + band forests in one launch (lm_k_pack_band_logits) vs the two launches: labelling launch alone, tests
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4t; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "label or stream or 4k or full_size or logits" > $O/gpu_tests_label.txt 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/gpu_tests_label.txt
[ $rc -eq 0 ] || { grep -n "Error\|assert" $O/gpu_tests_label.txt | head; exit 1; }
for F in 0 1 0 1; do
  echo "== LM_LABEL_FUSED_PACK=$F 1080p"
  LM_LABEL_FUSED_PACK=$F LM_LABEL_PARTS=1 timeout -k 10 120 python tools/label_microbench.py 64 1080 1920 5000 2>&1 | grep "FUSED"
done
for F in 0 1; do
  echo "== LM_LABEL_FUSED_PACK=$F 4K"
  LM_LABEL_FUSED_PACK=$F LM_LABEL_PARTS=1 timeout -k 10 120 python tools/label_microbench.py 16 2160 3840 5000 2>&1 | grep "FUSED"
done
for F in 0 1; do
  echo "== LM_LABEL_FUSED_PACK=$F 1080p sparse start of the stream"
  LM_LABEL_FUSED_PACK=$F LM_LABEL_PARTS=1 timeout -k 10 120 python tools/label_microbench.py 64 1080 1920 192 2>&1 | grep "FUSED"
done
