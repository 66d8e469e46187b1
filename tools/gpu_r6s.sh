#!/bin/bash
# round 6, call d2: the dense head's optimizer-only work behind the BPTT chain (wide layers, single-call step) -- tests, C4 A/B, timeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_parity.py tests/test_reference_layers.py tests/test_gpu_round5.py -m gpu -q -k "not c5_as_benched" > $out/r6s_tests.txt 2>&1; tail -3 $out/r6s_tests.txt
P=SBR_LIB=tools/probes/variants/libsbr_predefer.so
tools/gpu_call.sh r6s "ab:c4:$P:X=1:$P:X=2" "timeline:c4"
sed -n 1,40p $out/r6s_c4_timeline.txt | cut -c1-140
