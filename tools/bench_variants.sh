#!/bin/bash
# Runs ON THE GPU BOX: tools/bench_variants.sh <tag> "VAR=val VAR2=val" "..." : one bench.py line per environment variant
tag=$1; shift
mkdir -p gpurun_out
i=0
for v in "$@"; do
  i=$((i+1))
  env $v timeout 120 python bench.py --no-cpu-baseline --repeats 3 > gpurun_out/${tag}_v$i.json 2> gpurun_out/${tag}_v$i.err
  python - <<P
import json
try:
    d=json.loads(open("gpurun_out/${tag}_v$i.json").read().strip().splitlines()[-1])
    print("$v".ljust(44), d["ms_per_step"], {k: d["phases_us"][k] for k in ("rec_fwd","output","rec_bwd","wgrad","scatter","update")})
except Exception as e:
    print("$v", "FAILED", e); print(open("gpurun_out/${tag}_v$i.err").read()[-600:])
P
done
