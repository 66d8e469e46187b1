#!/bin/bash
# round 6, call e2: plain sorts of 8 k .. 30 k ids (C4: 26 744) on the hash-grouped counting kernels instead of the LDS histogram -- tests, C4 A/B, timeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_parity.py tests/test_gpu_wide_scatter_forms.py tests/test_gpu_round5.py -m gpu -q -k "not c5_as_benched" > $out/r6s_tests.txt 2>&1; tail -3 $out/r6s_tests.txt
P=SBR_LIB=tools/probes/variants/libsbr_presort2.so
tools/gpu_call.sh r6s "ab:c4:$P:X=1:$P:X=2" "timeline:c4"
grep -i "scat_\|softmax\|rec_bwd" $out/r6s_c4_timeline.txt | head -12 | cut -c1-130
