#!/bin/bash
# round 3, GPU call n: stamps of the tail's consumers (tools/tail_trace.py) for the tail-v3 default and with its parts off
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in "SBR_X=1" "SBR_TAIL_FENCE_KB=0 SBR_TAIL_EARLY_SORT=0 SBR_TAIL_OUT_STREAM=0" "SBR_TAIL_OUT_STREAM=0" "SBR_TAIL_PUBLISH_EVERY=1"; do
  echo "=== $v"
  env $v timeout 120 python tools/tail_trace.py 8 2>&1 | tail -45
done > gpurun_out/r3n_trace.txt 2>&1
cat gpurun_out/r3n_trace.txt
