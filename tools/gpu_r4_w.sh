#!/bin/bash
# round 4, job W: HBM traffic of one FCN forward pass (separate FETCH_SIZE / WRITE_SIZE passes)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4w; mkdir -p $O; R=$GRAFT_REPO_ROOT
cd /tmp
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -- python3 $R/tools/fcn_microbench.py mixed 4 > $O/f.log 2>&1 || { tail -5 $O/f.log; exit 1; }
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -- python3 $R/tools/fcn_microbench.py mixed 4 > $O/w.log 2>&1 || { tail -5 $O/w.log; exit 1; }
cd $R
python3 tools/fcn_traffic_pmc.py $(ls $O/f/*/*counter_collection.csv | head -1) $(ls $O/w/*/*counter_collection.csv | head -1) $O/r04_fcn_traffic_pmc.json
rm -rf $O/f $O/w
