#!/bin/bash
# labelling launch profiled as ONE part per batch (LM_LABEL_PARTS=1) so that per-kernel averages are per 64-frame launch
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/final2; mkdir -p $O
export LM_LABEL_PARTS=1
python tools/label_microbench.py 64 1080 1920 5000 > $O/r02_label_microbench_parts1.txt 2>&1
python tools/label_microbench.py 64 1080 1920 192 >> $O/r02_label_microbench_parts1.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/p_label -o l -- python3 $GRAFT_REPO_ROOT/tools/label_microbench.py 64 1080 1920 5000 > $O/p_label.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $GRAFT_REPO_ROOT/tools/label_microbench.py 64 1080 1920 5000 > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $GRAFT_REPO_ROOT/tools/label_microbench.py 64 1080 1920 5000 > $O/write.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/db_to_stats_csv.py $(find $O/p_label -name "*_results.db" | head -1) $O/r02_final_label_microbench_kernel_stats.csv
python3 tools/pmc_traffic.py $(ls $O/fetch/*/*counter_collection.csv | head -1) $(ls $O/write/*/*counter_collection.csv | head -1) $O/r02_label_traffic_pmc.json 64
rm -rf $O/p_label $O/fetch $O/write
grep labels $O/r02_label_microbench_parts1.txt
