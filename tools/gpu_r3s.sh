#!/bin/bash
# round 3, GPU call s: stamps inside the polling GEMM workgroups
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in "SBR_X=1" "SBR_TAIL_SCATTER_WGS=1"; do
  echo "=== $v"
  env $v timeout 120 python tools/tail_trace.py 8 2>&1 | tail -140
done > gpurun_out/r3s_trace.txt 2>&1
sed -n "/N tile 0/,/scatter-add/p" gpurun_out/r3s_trace.txt
