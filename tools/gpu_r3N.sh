#!/bin/bash
# round 3, GPU call N: the monitor as a workgroup of the scatter-add launch -- plain step and data-parallel step (one rank, RCCL)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_parity.py tests/test_gpu_dp_two_ranks.py -m gpu -x -q -k "overlapped or c2 or tail or ranks" 2>&1 | tail -3
run() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --sustained-seconds 0 --loop-iters 0 --repeats 3 $2 > gpurun_out/r3N.json 2> gpurun_out/r3N.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r3N.json").read().strip().splitlines()[-1])
dp=d.get("data_parallel") or {}
print("$1 $2".ljust(60), d["ms_per_step"], dp.get("exposed_us_per_step"))
P
}
( run "SBR_X=1" ""; run "SBR_X=1" "--force-dp"; run "SBR_TAIL_MONITOR_IN_UNITS=0" "--force-dp"; run "SBR_TAIL_OVERLAP=0" "--force-dp"; run "SBR_TAIL_MONITOR_IN_UNITS=0" "" ) 2>&1 | tee gpurun_out/r3N_dp.txt
