#!/bin/bash
# round 6, call q: the wide-tile GEMM -- parity tests, then C5 on the bf16 arithmetic with the previous library and this one (same box)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_gemm.py tests/test_gpu_layer_gemms.py -m gpu -q > $out/r6q_gemm_tests.txt 2>&1; tail -4 $out/r6q_gemm_tests.txt
tools/gpu_call.sh r6q2 "ab:c5:SBR_LIB=tools/probes/variants/libsbr_pregate.so,SBR_BENCH_FLAGS=384:SBR_BENCH_FLAGS=384" "ab:c4:SBR_LIB=tools/probes/variants/libsbr_pregate.so:X=1"
