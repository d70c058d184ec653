#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3o; mkdir -p $O
cd /tmp
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $O/a -- python3 $GRAFT_REPO_ROOT/tools/chain_profile.py 1280 > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 tools/kernel_pmc.py $(ls $O/a/*/*counter_collection.csv | head -1) lm_k_emit lm_k_stats lm_k_mb_twin_cmp lm_k_render_frames lm_k_mb_resolve lm_k_select lm_k_mb_tempo lm_k_band lm_k_write_labels lm_k_pack_rows_logits | tee $O/kernel_pmc.txt
rm -rf $O/a
