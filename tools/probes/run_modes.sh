#!/bin/bash
# bench phases for a list of SBR_X6_PIPE values
for m in "$@"; do
  r=$(SBR_X6_PIPE=$m timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['phases_us']['rec_fwd'], d['phases_us']['rec_bwd'])")
  echo "SBR_X6_PIPE=$m: seq/s, rec_fwd, rec_bwd us: $r"
done
