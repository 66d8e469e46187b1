#!/bin/bash
# round 3, GPU call I: ragged batches (ML-1M-shaped lengths): step and training loop with the rebuilt tail / without the overlapped tail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in "SBR_X=1" "SBR_TAIL_OVERLAP=0" "SBR_TAIL_SCATTER_LDS=0" "SBR_TAIL_FIRST=1 SBR_TAIL_GEOM=2.6"; do
  echo "=== $v"
  env $v timeout 200 python bench.py --lengths ml1m --quick --no-cpu-baseline --repeats 2 2>gpurun_out/r3I.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('phases_us'))"
  grep -i "error\|fault\|gave up\|skipped" gpurun_out/r3I.err | head -3
  env $v timeout 200 python tools/bench_train_loop.py --iters 1000 --host-iters 1 2>&1 | grep -i "metric\|error\|gave up" | cut -c150-330
done
