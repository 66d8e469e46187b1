#!/bin/bash
# round 3, GPU call u: when the scatter-add's waves get to their pieces
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in "SBR_X=1" "SBR_TAIL_SLAB_GROWTH=0.8" "SBR_TAIL_SLAB_MAX=512" "SBR_TAIL_FENCE_KB=0"; do
  echo "=== $v"
  env $v timeout 120 python tools/tail_trace.py 8 2>&1 | tail -160
done > gpurun_out/r3u_trace.txt 2>&1
grep -A8 "^===\|^scatter-add\|ends per" gpurun_out/r3u_trace.txt | grep -v "^    [+-]\|^   *[0-9]*:" | cut -c1-150
