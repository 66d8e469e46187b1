#!/bin/bash
# round 6, call g: is the Hp = 512 step slow because two workgroups share a CU?  (C5's chains at 128 rows = one workgroup per CU); scalar vs packed gate math
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
for b in 256 128; do timeout 300 python tools/cl_prof.py c5 $b > $out/r6g_cluster_phases_c5_b$b.txt 2>&1; cat $out/r6g_cluster_phases_c5_b$b.txt | cut -c1-420; done
tools/gpu_call.sh r6g "ab:c4:X=1:SBR_LIB=tools/probes/variants/libsbr_noslp.so" "ab:c5:SBR_SCAT_RANGE_MAX=2048:SBR_SCAT_RANGE_MAX=2048,SBR_BENCH_FLAGS=384"
timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -x -k "head" > $out/r6g_tests_1.txt 2>&1; tail -3 $out/r6g_tests_1.txt | cut -c1-300
