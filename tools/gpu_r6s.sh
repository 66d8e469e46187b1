#!/bin/bash
# round 6, call b2: the row-sparse step skips list entries that repeat their predecessor -- tests, C3 / C5 A/B, C3 timeline
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
timeout 2400 python -m pytest tests/test_gpu_sparse_update.py tests/test_gpu_config_parity.py tests/test_gpu_parity.py -m gpu -q -k "not c5_as_benched" > $out/r6s_tests.txt 2>&1; tail -3 $out/r6s_tests.txt
P=SBR_LIB=tools/probes/variants/libsbr_preskip.so
tools/gpu_call.sh r6s "ab:c3:$P:X=1:$P:X=2" "ab:c5:$P:X=1" "timeline:c3"
grep -i "sp_rows" $out/r6s_c3_timeline.txt | head -4 | cut -c1-120
