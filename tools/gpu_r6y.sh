#!/bin/bash
# round 6, call y (sixth run): the combining scatter-add for wide rows (C5: 2 048 floats) -- tests, C5 A/B, timeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
timeout 1800 python -m pytest tests/test_gpu_wide_scatter_forms.py tests/test_gpu_config_parity.py tests/test_gpu_parity.py -m gpu -q -k "wide or scatter or c5 or two_layer or 512" > $out/r6y_tests.txt 2>&1; tail -3 $out/r6y_tests.txt
tools/gpu_call.sh r6y "ab:c5:SBR_LIB=tools/probes/variants/libsbr_prec5s.so:X=1" "timeline:c5"
grep -i "scat_reduce" $out/r6y_c5_timeline.txt | head -2 | cut -c1-130
