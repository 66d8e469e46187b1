#!/bin/bash
# round 3, GPU call W: the driver's bench line on the tree as committed
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 100 python bench.py > gpurun_out/r3W_bench.json 2> gpurun_out/r3W_bench.err
python -c "
import json; d=json.loads(open('gpurun_out/r3W_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value','ms_per_step')}, d['sustained']['ms_per_step'], d.get('train_loop',{}).get('ms_per_iteration'), d.get('train_loop_error'), d['roofline']['frac'], d['roofline']['traffic'], d['cpu_baseline']['value'])"
