#!/bin/bash
# round 3, GPU call v: polling GEMM as persistent groups over the slab table
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_parity.py -m gpu -x -q -k "overlapped or c2 or tail or pipelined" 2>&1 | tail -5
for v in "SBR_X=1" "SBR_TAIL_GEMM_GROUPS=48" "SBR_TAIL_GEMM_GROUPS=76"; do
  echo "=== $v"
  env $v timeout 120 python tools/tail_trace.py 8 2>&1 | tail -160
done > gpurun_out/r3v_trace.txt 2>&1
grep -v "^    [+-]" gpurun_out/r3v_trace.txt | cut -c1-130 | awk '/^   *[0-9]+:/ { if ((n++ % 3) == 0) print; next } { print }'
tools/bench_variants.sh r3v "SBR_DUMMY=1" "SBR_TAIL_GEMM_GROUPS=48" "SBR_TAIL_GEMM_GROUPS=76" "SBR_TAIL_FENCE_KB=0 SBR_TAIL_EARLY_SORT=0 SBR_TAIL_OUT_STREAM=0" "SBR_TAIL_OUT_STREAM=0" "SBR_TAIL_SLAB_GROWTH=0.25" "SBR_DUMMY=2" 2>&1 | tee gpurun_out/r3v_variants.txt
