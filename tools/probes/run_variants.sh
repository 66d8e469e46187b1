#!/bin/bash
# rec_fwd time of each experiment variant (see x6p_variants.sh), gated (SBR_X6_PIPE=2) and ungated (1)
for n in "$@"; do
  for p in 2 1; do
    lib=tools/probes/variants/libsbr_dbg$n.so; [ "$n" = 0 ] && lib=sequence-based-recommendations_amd/libsbr_rnn.so
    r=$(SBR_LIB=$lib SBR_X6_PIPE=$p timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['phases_us']['rec_fwd'])" 2>&1 | tail -1)
    echo "dbg=$n pipe=$p rec_fwd_us=$r"
  done
done
