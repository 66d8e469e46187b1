#!/bin/bash
# round 6, call u: the index-input block's row step in front of the side-stream join -- sparse tests, C3 / C5 A/B against the commit before
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_sparse_update.py tests/test_gpu_config_parity.py -m gpu -q -k "not c5_as_benched" > $out/r6u_tests.txt 2>&1; tail -3 $out/r6u_tests.txt
tools/gpu_call.sh r6u "ab:c3:SBR_LIB=tools/probes/variants/libsbr_prerow.so:X=1:SBR_LIB=tools/probes/variants/libsbr_prerow.so:X=2" "ab:c5:SBR_LIB=tools/probes/variants/libsbr_prerow.so:X=1"
