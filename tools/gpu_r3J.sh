#!/bin/bash
# round 3, GPU call J: the driver's bench line (all legs) on the rebuilt tail
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python bench.py > gpurun_out/r3J_bench.json 2> gpurun_out/r3J_bench.err
grep -i "train loop\|skipped\|error\|gave up\|Traceback" gpurun_out/r3J_bench.err | head -10
python - <<P
import json
d=json.loads(open("gpurun_out/r3J_bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("sustained"), d.get("train_loop"), d["roofline"]["frac"], d["roofline"].get("traffic"), d["phases_us"])
P
