#!/bin/bash
# round 6, call f: ADVICE fixes (tests), batched merge of the range scatter-add, the untouched rows' step beside the BPTT chain again
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
V=SBR_LIB=tools/probes/variants/libsbr_r6c.so
timeout 1200 python -m pytest tests/test_gpu_round5.py tests/test_gpu_parity.py tests/test_gpu_dp_two_ranks.py -m gpu -q -x -k "head or tail or two_ranks or wide or sampled" > $out/r6f_tests_1.txt 2>&1; tail -5 $out/r6f_tests_1.txt | cut -c1-300
tools/gpu_call.sh r6f "ab:c4:X=1:$V:SBR_EARLY_UPDATE=1:SBR_EARLY_UPDATE=2" "ab:c4:SBR_EARLY_UPDATE=2,SBR_UNTOUCHED_WGS=256:SBR_EARLY_UPDATE=2,SBR_UNTOUCHED_WGS=64" "timeline:c4:SBR_EARLY_UPDATE=2,SBR_UNTOUCHED_WGS=256"
