#!/bin/bash
# Regenerates the stream-side profiles/r01_final_* artifacts on a GPU box (run from the repo root through gpurun); results land in
# gpurun_out/final/ and are copied into profiles/ by hand.  rocprofv3 gets the program itself after "--" (no wrappers).
set -e
R=$(pwd)
O=$R/gpurun_out/final
mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench.log 2>&1 && grep '^{"metric' $O/bench.log > $O/r01_final_bench.json
python bench.py --no-pipeline > $O/bench_np.log 2>&1 && grep '^{"metric' $O/bench_np.log > $O/r01_final_bench_nopipeline.json
python tools/label_microbench.py 64 > $O/r01_final_label_microbench.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --stats -d $O/p_bench -- python3 $R/bench.py --steps 6 --cpu-frames 0 > $O/p_bench.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/p_bench_np -- python3 $R/bench.py --steps 6 --cpu-frames 0 --no-pipeline > $O/p_bench_np.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/p_label -- python3 $R/tools/label_microbench.py 64 > $O/p_label.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/p_fetch -- python3 $R/tools/label_microbench.py 64 > $O/p_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/p_write -- python3 $R/tools/label_microbench.py 64 > $O/p_write.log 2>&1
cd $R
python tools/db_to_stats_csv.py $(ls $O/p_bench/*/*_results.db | head -1) $O/r01_final_bench_kernel_stats.csv
python tools/db_to_stats_csv.py $(ls $O/p_bench_np/*/*_results.db | head -1) $O/r01_final_bench_nopipeline_kernel_stats.csv
python tools/db_to_stats_csv.py $(ls $O/p_label/*/*_results.db | head -1) $O/r01_final_label_microbench_kernel_stats.csv
python tools/prof_overlap.py $(ls $O/p_bench/*/*_results.db | head -1) > $O/r01_final_bench_overlap.txt
python tools/pmc_traffic.py $(ls $O/p_fetch/*/*counter_collection.csv | head -1) $(ls $O/p_write/*/*counter_collection.csv | head -1) $O/r01_label_traffic_pmc.json 64
rm -rf $O/p_bench $O/p_bench_np $O/p_label        # the databases are large; the CSV summaries are what is kept
echo refreshed
