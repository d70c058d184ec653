#!/bin/bash
# PMC passes (counters only, no trace domains): HBM traffic of the labelling sequence, MFMA busy cycles of the FCN
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/pmc; mkdir -p $O
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $GRAFT_REPO_ROOT/tools/label_microbench.py 64 1080 1920 5000 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $GRAFT_REPO_ROOT/tools/label_microbench.py 64 1080 1920 5000 > $O/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/mfma -- python3 $GRAFT_REPO_ROOT/bench.py --workload fcn --steps 2 --warmup 1 --no-fcn-oracle > $O/mfma.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_traffic.py $(ls $O/fetch/*/*counter_collection.csv | head -1) $(ls $O/write/*/*counter_collection.csv | head -1) $O/r02_label_traffic_pmc.json 64
python3 tools/fcn_mfma_pmc.py $(ls $O/mfma/*/*counter_collection.csv | head -1) $O/r02_fcn_mfma_pmc_f16x3.json
rm -rf $O/fetch $O/write $O/mfma
