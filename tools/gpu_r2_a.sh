#!/bin/bash
# round 2, call A: the 10k-frame stream for the first time + the N=2 rehearsal of the sharded path
set -x
mkdir -p gpurun_out
cd "$GRAFT_REPO_ROOT"
LM_BENCH_VERBOSE=1 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --fcn-frames 3 --e2e-frames 16 > gpurun_out/a_bench.json 2> gpurun_out/a_bench.err
echo "bench rc=$?"
tail -c 3000 gpurun_out/a_bench.err
cat gpurun_out/a_bench.json
timeout -k 10 200 python bench.py --frames 1024 --steps 1 --warmup 1 --cpu-frames 0 --fcn-frames 0 > gpurun_out/a_n1_1024.json 2> gpurun_out/a_n1_1024.err
echo "n1 rc=$?"
LM_BENCH_REHEARSE=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --frames 1024 --steps 1 --warmup 1 --cpu-frames 0 --fcn-frames 0 > gpurun_out/a_n2_1024.json 2> gpurun_out/a_n2_1024.err
echo "n2 rc=$?"
tail -c 1500 gpurun_out/a_n2_1024.err
python - <<'PY'
import json
a=json.loads(open('gpurun_out/a_n1_1024.json').read().strip().splitlines()[-1])
b=json.loads(open('gpurun_out/a_n2_1024.json').read().strip().splitlines()[-1])
print("N1", a["value"], a["parity"]["digests"])
print("N2", b["value"], b["parity"]["digests"])
print("digests equal:", a["parity"]["digests"]==b["parity"]["digests"])
PY
