#!/bin/bash
# round 3, GPU call M: the data-parallel step (one rank, RCCL / gloo) on the rebuilt tail: overhead against the plain step
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for a in "" "--force-dp" "--force-dp --dp-backend gloo"; do
  timeout 300 python bench.py --no-cpu-baseline --no-pmc --sustained-seconds 0 --loop-iters 0 --repeats 3 $a > gpurun_out/r3M.json 2> gpurun_out/r3M.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r3M.json").read().strip().splitlines()[-1])
print("$a".ljust(30), d["ms_per_step"], d.get("data_parallel"))
P
done
timeout 300 python bench.py --gpus 2 --dp-backend gloo --no-cpu-baseline --no-pmc --sustained-seconds 0 --loop-iters 0 --repeats 3 2>gpurun_out/r3M2.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('--gpus 2 gloo', d['ms_per_step'], d['value'], d.get('data_parallel'))"
