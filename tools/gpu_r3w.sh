#!/bin/bash
# round 3, GPU call w: does the chain slow down because of the scatter-add's atomics?  (variant: no flush -- wrong gradients)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in "SBR_X=1" "SBR_LIB=tools/probes/variants/libsbr_noflush.so" "SBR_TAIL_SCATTER_WGS=1"; do
  echo "=== $v"
  env $v timeout 120 python tools/tail_trace.py 8 2>&1 | tail -160
done > gpurun_out/r3w_trace.txt 2>&1
grep "^===\|^chain\|^GEMM\|^scatter\|k=\|ends per" gpurun_out/r3w_trace.txt | cut -c1-150
