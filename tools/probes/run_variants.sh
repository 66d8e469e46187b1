#!/bin/bash
# rec_fwd / rec_bwd time of each experiment variant (see x6p_build.sh)
for n in "$@"; do
    lib=tools/probes/variants/libsbr_dbg$n.so; [ "$n" = 0 ] && lib=sequence-based-recommendations_amd/libsbr_rnn.so
    r=$(SBR_LIB=$lib timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['phases_us']['rec_fwd'], d['phases_us']['rec_bwd'])" 2>&1 | tail -1)
    echo "dbg=$n rec_fwd rec_bwd us: $r"
done
