#!/bin/bash
# round 3, GPU call h: x6r forward schedule 2 + x6r backward (first form)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
SBR_X6R_BWD=0 timeout 600 python -m pytest tests/test_gpu_config_parity.py tests/test_reference_layers.py tests/test_gpu_parity.py -m gpu -x -q -k "c2 or reference or one_layer or pipelined or overlapped or smallest or ragged" > gpurun_out/r3h_tests_fwd.txt 2>&1; tail -5 gpurun_out/r3h_tests_fwd.txt
timeout 900 python -m pytest tests/test_gpu_config_parity.py tests/test_reference_layers.py tests/test_gpu_parity.py tests/test_gpu_edge_shapes.py tests/test_gpu_dp_two_ranks.py -m gpu -x -q -k "c2 or c1 or reference or one_layer or pipelined or overlapped or fp16 or smallest or two_layer or ragged or unfused or bf16x6 or edge or ranks" > gpurun_out/r3h_tests_both.txt 2>&1; tail -12 gpurun_out/r3h_tests_both.txt
tools/bench_variants.sh r3h "SBR_X6R=0 SBR_X6R_BWD=0" "SBR_X6R=1 SBR_X6R_BWD=0" "SBR_X6R=0 SBR_X6R_BWD=1" "SBR_X6R=1 SBR_X6R_BWD=1" "SBR_X6R=0 SBR_X6R_BWD=0 SBR_TAIL_OVERLAP=0" "SBR_X6R=0 SBR_X6R_BWD=1 SBR_TAIL_OVERLAP=0" "SBR_X6R=0 SBR_X6R_BWD=0 SBR_Y=2" 2>&1 | tee gpurun_out/r3h_variants.txt
