#!/bin/bash
# round 3, GPU call K: 16-byte optimizer pass, fewer driver calls per step (function attributes once, no memset in front of the sort)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_parity.py tests/test_gpu_dp_two_ranks.py tests/test_gpu_train_cli.py -m gpu -x -q -k "overlapped or c2 or tail or pipelined or ranks or updater or learns or one_layer_cce or sampled" 2>&1 | tail -4
tools/bench_variants.sh r3K "SBR_DUMMY=1" "SBR_UPDATE_V4=0" "SBR_DUMMY=2" "SBR_UPDATE_V4=0 SBR_X=2" 2>&1 | tee gpurun_out/r3K_variants.txt
for v in "SBR_X=1" "SBR_TAIL_OVERLAP=0"; do env $v timeout 200 python tools/bench_train_loop.py --iters 1000 --host-iters 1 2>&1 | grep -i "metric\|error\|gave up" | cut -c150-330; done
