#!/bin/bash
# round 3, GPU call q: per-slab stamps of the polling GEMM
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in "SBR_X=1" "SBR_TAIL_SCATTER_WGS=1"; do
  echo "=== $v"
  env $v timeout 120 python tools/tail_trace.py 8 2>&1 | tail -140
done > gpurun_out/r3q_trace.txt 2>&1
cat gpurun_out/r3q_trace.txt
