#!/bin/bash
# round 6, call y: small layers (C1) -- the f32 weight-gradient kernel with up to eight k-steps in flight, 16 narrow rows in flight in the scatter-add
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_wide_scatter_forms.py -m gpu -q > $out/r6y_tests.txt 2>&1; tail -3 $out/r6y_tests.txt
tools/gpu_call.sh r6y "ab:c1:SBR_LIB=tools/probes/variants/libsbr_prec1.so:X=1:SBR_LIB=tools/probes/variants/libsbr_prec1.so:X=2" "timeline:c1"
grep -i "wgrad_kernel\|scat_reduce" $out/r6y_c1_timeline.txt | head -3 | cut -c1-120
