#!/bin/bash
# round 6, call q2: the small-layer chains (rec_*_x6q) with a step's inputs two steps ahead in two register sets -- parity tests, C1 A/B, timeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_reference_layers.py tests/test_gpu_config_parity.py -m gpu -q -k "not c5_as_benched and not c3_c4 and not c5_shape" > $out/r6s_tests.txt 2>&1; tail -3 $out/r6s_tests.txt
P=SBR_LIB=tools/probes/variants/libsbr_preq.so
tools/gpu_call.sh r6s "ab:c1:$P:X=1:$P:X=2" "timeline:c1"
grep -i "rec_fwd\|rec_bwd" $out/r6s_c1_timeline.txt | head -2 | cut -c1-120
