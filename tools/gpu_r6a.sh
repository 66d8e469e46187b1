#!/bin/bash
# round 6, call a: the rebuilt 16-row cluster chains (sbr_rec_c16.hip) -- parity subset, phase counters, same-box A/B against round 5's library
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
R5=SBR_LIB=tools/probes/variants/libsbr_r5.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cluster or width or wide or 512" > $out/r6a_tests_1.txt 2>&1; tail -8 $out/r6a_tests_1.txt | cut -c1-300
for c in c4 c5; do timeout 300 python tools/cl_prof.py $c > $out/r6a_cluster_phases_$c.txt 2>&1; cat $out/r6a_cluster_phases_$c.txt | cut -c1-330; done
tools/gpu_call.sh r6a "ab:c4:X=1:$R5" "ab:c3:X=1:$R5" "ab:c5:X=1:$R5"
timeout 1500 python -m pytest tests/test_gpu_config_parity.py -m gpu -q -k "c3 or c4 or c5" > $out/r6a_tests_2.txt 2>&1; tail -12 $out/r6a_tests_2.txt | cut -c1-300
