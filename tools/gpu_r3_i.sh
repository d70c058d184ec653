#!/bin/bash
# round 3, job I: counters of the planar FCN pass: MFMA-pipe utilisation, stalls, LDS conflicts
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3i; mkdir -p $O
timeout -k 10 200 python bench.py --workload fcn --steps 20 --warmup 3 --no-fcn-oracle > $O/fcn_bench.json 2> $O/fcn_bench.err || { tail -5 $O/fcn_bench.err; exit 1; }
python3 -c "
import json; d=json.load(open('gpurun_out/r3i/fcn_bench.json')); f=d['fcn']; print({k: f.get(k) for k in ('ms_per_frame','algorithmic_tflops','executed_gflop_per_frame','executed_tflops','frac_of_peak_executed')})"
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/m -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcn --steps 2 --warmup 1 --no-fcn-oracle > $O/m.log 2>&1 || { tail -5 $O/m.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU --output-format csv -d $O/a -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcn --steps 1 --warmup 1 --no-fcn-oracle > $O/a.log 2>&1 || { tail -5 $O/a.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY --output-format csv -d $O/b -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcn --steps 1 --warmup 1 --no-fcn-oracle > $O/b.log 2>&1 || { tail -5 $O/b.log; exit 1; }
cd $GRAFT_REPO_ROOT
python3 tools/fcn_mfma_pmc.py $(ls $O/m/*/*counter_collection.csv | head -1) $O/r03_fcn_mfma_pmc_mixed.json
python3 tools/fcn_stall_pmc.py $(ls $O/a/*/*counter_collection.csv | head -1) $O/r03_fcn_stalls_a.txt 30
python3 tools/fcn_stall_pmc.py $(ls $O/b/*/*counter_collection.csv | head -1) $O/r03_fcn_stalls_b.txt 30
rm -rf $O/m $O/a $O/b
