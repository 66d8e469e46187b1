#!/bin/bash
# round 6, call j: the Hp = 512 forward chain with 32 units per workgroup (one workgroup per CU)
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
SBR_C16_UT2=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "512 or eight_row" > $out/r6j_tests_1.txt 2>&1; tail -4 $out/r6j_tests_1.txt | cut -c1-300
SBR_C16_UT2=1 timeout 300 python tools/cl_prof.py c5 > $out/r6j_cluster_phases_c5.txt 2>&1; head -6 $out/r6j_cluster_phases_c5.txt | cut -c1-420
tools/gpu_call.sh r6j "ab:c5:SBR_C16_UT2=0:SBR_C16_UT2=1"
