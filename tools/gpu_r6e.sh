#!/bin/bash
# round 6, call e: range scatter-add pipeline with the flushes counted out of its waits
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
V=SBR_LIB=tools/probes/variants/libsbr_r6c.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round5.py tests/test_gpu_config_parity.py -m gpu -q -x -k "wide or c3 or c4 or scatter" > $out/r6e_tests_1.txt 2>&1; tail -5 $out/r6e_tests_1.txt | cut -c1-300
tools/gpu_call.sh r6e "ab:c4:X=1:$V" "ab:c3:X=1:$V" "ab:c5:X=1:SBR_SCAT_RANGE_MAX=2048" "timeline:c4"
