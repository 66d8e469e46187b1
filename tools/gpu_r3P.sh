#!/bin/bash
# round 3, GPU call P: final tree -- data-parallel step with one collective (default) / three, the bench line, the whole GPU suite
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
run() { env $1 timeout 300 python bench.py --no-cpu-baseline --no-pmc --sustained-seconds 0 --loop-iters 0 --repeats 3 $2 > gpurun_out/r3P.json 2> gpurun_out/r3P.err
  python - <<P
import json
d=json.loads(open("gpurun_out/r3P.json").read().strip().splitlines()[-1])
dp=d.get("data_parallel") or {}
print("$1 $2".ljust(50), d["ms_per_step"], dp.get("collectives_per_step"), dp.get("exposed_us_per_step"))
P
}
( run "SBR_X=1" ""; run "SBR_X=1" "--force-dp"; run "SBR_DP_OVERLAP=1" "--force-dp"; run "SBR_X=1" "--force-dp --dp-backend gloo" ) 2>&1 | tee gpurun_out/r3P_dp.txt
timeout 600 python bench.py > gpurun_out/r3P_bench.json 2> gpurun_out/r3P_bench.err
python -c "
import json; d=json.loads(open('gpurun_out/r3P_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step')}, d['sustained']['ms_per_step'], d.get('train_loop',{}).get('ms_per_iteration'), d['roofline']['frac'], d['roofline']['traffic'])"
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/r3P_gputests.txt 2>&1
tail -8 gpurun_out/r3P_gputests.txt | cut -c1-250
