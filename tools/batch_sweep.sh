#!/bin/bash
# throughput of the default config vs rows per GPU:  tools/batch_sweep.sh [batches...]
for b in ${@:-128 256 512 1024 2048}; do
  python bench.py --batch $b --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    try: d = json.loads(l)
    except Exception: continue
    p = d['phases_us']
    print('B=%5s  %9.1f seq/s  %.3f ms/step  rec_fwd %.0f  rec_bwd %.0f us' % (sys.argv[1], d['value'], d['ms_per_step'], p['rec_fwd'], p['rec_bwd']))
" $b
done
