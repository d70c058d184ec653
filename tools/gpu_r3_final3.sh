#!/bin/bash
# round 3, session 2: configs[4] line (4K stream + 4K e2e) and the two-rank rehearsal on one GPU, with the final library
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3final3; mkdir -p $O
timeout -k 10 500 python bench.py --gpus 1 --steps 5 --warmup 2 --height 2160 --width 3840 --frames 1024 --batch 16 --cpu-frames 0 --fcn-frames 3 --e2e-frames 32 --no-fcn-oracle > $O/r03_final_bench_4k.json 2> $O/bench4k.err || { tail -20 $O/bench4k.err; exit 1; }
python3 - $O/r03_final_bench_4k.json <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print("4K value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "alone", d["roofline"]["alone"]["frac"], "e2e", d.get("e2e_rgb", {}).get("value"))
PY
LM_BENCH_REHEARSE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 2 --warmup 1 --fcn-frames 0 --cpu-frames 0 > $O/r03_final_bench_rehearse_n2.json 2> $O/rehearse.err || { tail -30 $O/rehearse.err; exit 1; }
python3 - $O/r03_final_bench_rehearse_n2.json <<'PY'
import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("rehearse n=2 value", d["value"], "parity", d["parity"]["match"], "scaling", d["scaling"], {k: d.get(k) for k in ("amdahl",)})
PY
