#!/bin/bash
# round 4, job J: rocprof evidence -- labelling launch at 1080p and at 4K (kernel stats, HBM traffic from separate FETCH / WRITE passes),
# MFMA-pipe counters of the FCN pass (1080p = the size the network also runs at for 4K frames), the 4K bench line with its parity
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r4j; mkdir -p $O
R=$GRAFT_REPO_ROOT
label() { # tag batch h w
  cd /tmp
  LM_LABEL_PARTS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/p_$1 -o l -- python3 $R/tools/label_microbench.py $2 $3 $4 5000 > $O/${1}_label_microbench.txt 2>&1 || { tail -5 $O/${1}_label_microbench.txt; exit 1; }
  LM_LABEL_PARTS=1 timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f_$1 -- python3 $R/tools/label_microbench.py $2 $3 $4 5000 > $O/f_$1.log 2>&1 || { tail -5 $O/f_$1.log; exit 1; }
  LM_LABEL_PARTS=1 timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w_$1 -- python3 $R/tools/label_microbench.py $2 $3 $4 5000 > $O/w_$1.log 2>&1 || { tail -5 $O/w_$1.log; exit 1; }
  cd $R
  python3 tools/db_to_stats_csv.py $(find $O/p_$1 -name "*_results.db" | head -1) $O/${1}_label_microbench_kernel_stats.csv
  python3 tools/pmc_traffic.py $(ls $O/f_$1/*/*counter_collection.csv | head -1) $(ls $O/w_$1/*/*counter_collection.csv | head -1) $O/${1}_label_traffic_pmc.json $2 $4 $3
  rm -rf $O/p_$1 $O/f_$1 $O/w_$1
  grep -v "amdgpu.ids\|rocprofv3" $O/${1}_label_microbench.txt | tail -3; head -8 $O/${1}_label_microbench_kernel_stats.csv | cut -c1-40,120-; python3 -c "import json; d=json.load(open('$O/${1}_label_traffic_pmc.json')); print({k: d[k] for k in d if 'bytes' in k or 'ratio' in k})"
}
label r04 64 1080 1920
label r04_4k 16 2160 3840
cd /tmp
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/m -- python3 $R/bench.py --workload fcn --height 2160 --width 3840 --steps 2 --warmup 1 --no-fcn-oracle > $O/m.log 2>&1 || { tail -5 $O/m.log; exit 1; }
cd $R
python3 tools/fcn_mfma_pmc.py $(ls $O/m/*/*counter_collection.csv | head -1) $O/r04_4k_fcn_mfma_pmc_mixed.json | tail -30
rm -rf $O/m
timeout -k 10 600 python bench.py --height 2160 --width 3840 --frames 1024 --batch 16 --steps 5 --warmup 4 --fcn-frames 20 --e2e-frames 32 > $O/r04_4k_bench.json 2> $O/b4k.err || { tail -20 $O/b4k.err; exit 1; }
python3 - $O/r04_4k_bench.json <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print("4K value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"], d["parity"]["reference"])
print("roofline", {k: d["roofline"][k] for k in ("frac","launch_ms","alone","traffic")})
print("fcn", {k: d["fcn"].get(k) for k in ("ms_per_frame","frac_of_peak_algorithmic")}, "e2e", d["e2e_rgb"].get("value"))
PY
