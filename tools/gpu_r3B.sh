#!/bin/bash
# round 3, GPU call B: monitor on the third side stream, output-layer kernels on the scatter stream in front of the units
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_parity.py tests/test_gpu_dp_two_ranks.py -m gpu -x -q -k "overlapped or c2 or tail or pipelined or ranks" 2>&1 | tail -5
for v in "SBR_X=1" "SBR_TAIL_OUT_STREAM=0"; do
  echo "=== $v"
  env $v timeout 120 python tools/tail_trace.py 8 2>&1 | tail -160
done > gpurun_out/r3B_trace.txt 2>&1
grep -v "^    [+-]" gpurun_out/r3B_trace.txt | cut -c1-150 | awk '/^   *[0-9]+:/ { if ((n++ % 12) == 0) print; next } { print }'
tools/bench_variants.sh r3B "SBR_DUMMY=1" "SBR_TAIL_OUT_STREAM=0" "SBR_TAIL_SCATTER_UNITS=256" "SBR_TAIL_GEMM_GROUPS=48" "SBR_TAIL_SLAB_MAX=1024" "SBR_TAIL_SCATTER_LDS=0" "SBR_TAIL_FENCE_KB=0" "SBR_DUMMY=2" 2>&1 | tee gpurun_out/r3B_variants.txt
