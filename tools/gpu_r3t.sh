#!/bin/bash
# round 3, GPU call t: polling GEMM with the transposed 16-byte epilogue and the slab table
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_parity.py tests/test_gpu_gemm.py -m gpu -x -q 2>&1 | tail -5
for v in "SBR_X=1" "SBR_TAIL_SLAB_GROWTH=0.35" "SBR_TAIL_SLAB_GROWTH=0.8"; do
  echo "=== $v"
  env $v timeout 120 python tools/tail_trace.py 8 2>&1 | tail -140
done > gpurun_out/r3t_trace.txt 2>&1
grep -v "^    [+-]" gpurun_out/r3t_trace.txt | cut -c1-120
tools/bench_variants.sh r3t "SBR_DUMMY=1" "SBR_TAIL_SLAB_GROWTH=0.35" "SBR_TAIL_SLAB_GROWTH=0.8" "SBR_TAIL_FENCE_KB=0 SBR_TAIL_EARLY_SORT=0 SBR_TAIL_OUT_STREAM=0" "SBR_TAIL_OUT_STREAM=0" "SBR_DUMMY=2" 2>&1 | tee gpurun_out/r3t_variants.txt
