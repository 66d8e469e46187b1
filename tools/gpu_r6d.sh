#!/bin/bash
# round 6, call d: the range scatter-add as a software pipeline + batched merge
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
V=SBR_LIB=tools/probes/variants/libsbr_r6c.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_gpu_config_parity.py -m gpu -q -x -k "cluster or width or wide or c3 or c4 or scatter" > $out/r6d_tests_1.txt 2>&1; tail -8 $out/r6d_tests_1.txt | cut -c1-300
tools/gpu_call.sh r6d "ab:c4:X=1:$V" "ab:c3:X=1:$V" "ab:c5:X=1:SBR_SCAT_RANGE_MAX=2048" "timeline:c4" "timeline:c3"
