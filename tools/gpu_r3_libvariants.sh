#!/bin/bash
# bench with library variants (tools/variants/lib_<name>.so copied over the in-tree library of this snapshot; "default" = as shipped);
# the first configuration is run twice (lease warm-up), every configuration's last run is printed
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/$1; shift; mkdir -p $O
cp lecturemath_amd/liblecturemath_hip.so /tmp/lib_default.so
first=1
for v in "$@"; do
  if [ "$v" = default ]; then cp /tmp/lib_default.so lecturemath_amd/liblecturemath_hip.so; else cp tools/variants/lib_$v.so lecturemath_amd/liblecturemath_hip.so; fi
  for rep in $(seq 1 $((first + 2))); do
  timeout -k 10 400 python bench.py --gpus 1 --steps 9 --warmup 3 --fcn-frames 0 --cpu-frames 0 > $O/bench_$v.json 2> $O/bench_$v.err || { tail -20 $O/bench_$v.err; exit 1; }
  python3 - $O/bench_$v.json $v $rep <<'PY'
import json, sys
d=json.load(open(sys.argv[1]))
print("variant", sys.argv[2], "rep", sys.argv[3], "value", d["value"], "ms/step", d["ms_per_step"], "parity", d["parity"]["match"], "frac", d["roofline"]["frac"])
PY
  done
  first=0
done
cp /tmp/lib_default.so lecturemath_amd/liblecturemath_hip.so
