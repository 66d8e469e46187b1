#!/bin/bash
# round 3, GPU call o/p: the monitor word replicated (o); consumers without the acquire fence, sc1 loads (p) -- stamps and step times
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in "SBR_X=1" "SBR_TAIL_FENCE_KB=0 SBR_TAIL_EARLY_SORT=0 SBR_TAIL_OUT_STREAM=0" "SBR_TAIL_SMALL_K=64 SBR_TAIL_SMALL_SLABS=128"; do
  echo "=== $v"
  env $v timeout 120 python tools/tail_trace.py 8 2>&1 | tail -45
done > gpurun_out/r3p_trace.txt 2>&1
cat gpurun_out/r3p_trace.txt
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_parity.py -m gpu -x -q -k "overlapped or c2" 2>&1 | tail -3
tools/bench_variants.sh r3p "SBR_DUMMY=1" "SBR_TAIL_FENCE_KB=0 SBR_TAIL_EARLY_SORT=0 SBR_TAIL_OUT_STREAM=0" "SBR_TAIL_OUT_STREAM=0" "SBR_TAIL_FENCE_KB=0" "SBR_DUMMY=2" 2>&1 | tee gpurun_out/r3p_variants.txt
