#!/bin/bash
# round 6, call b: the two-level backward exchange at Hp = 512 (rec_bwd_c16t)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "512 or eight_row or bf16x6_products or active_clip" > $out/r6b_tests_1.txt 2>&1; tail -8 $out/r6b_tests_1.txt | cut -c1-300
timeout 300 python tools/cl_prof.py c5 > $out/r6b_cluster_phases_c5.txt 2>&1; cat $out/r6b_cluster_phases_c5.txt | cut -c1-400
tools/gpu_call.sh r6b "ab:c5:SBR_C16_TWO_LEVEL=1:SBR_C16_TWO_LEVEL=0"
timeout 1500 python -m pytest tests/test_gpu_config_parity.py -m gpu -q -k "c5" > $out/r6b_tests_2.txt 2>&1; tail -12 $out/r6b_tests_2.txt | cut -c1-300
