#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/fcnpmc; mkdir -p $O
cd /tmp
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/a -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcn --steps 1 --warmup 1 --no-fcn-oracle > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --output-format csv -d $O/b -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcn --steps 1 --warmup 1 --no-fcn-oracle > $O/b.log 2>&1 || { tail -5 $O/b.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 tools/fcn_stall_pmc.py $(ls $O/a/*/*counter_collection.csv | head -1) $O/r02_fcn_stalls_a.txt 26
python3 tools/fcn_stall_pmc.py $(ls $O/b/*/*counter_collection.csv | head -1) $O/r02_fcn_stalls_b.txt 26
head -3 $(ls $O/b/*/*counter_collection.csv | head -1)
rm -rf $O/a $O/b
