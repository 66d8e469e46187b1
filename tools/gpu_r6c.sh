#!/bin/bash
# round 6, call c: K-contiguous gate packing (no DPP pairing) + requests behind the reduce barrier in the backward chains
cd ${GRAFT_REPO_ROOT:-/root/repo}
out=gpurun_out; mkdir -p $out
V=SBR_LIB=tools/probes/variants/libsbr_r6b.so
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cluster or width or wide or 512" > $out/r6c_tests_1.txt 2>&1; tail -8 $out/r6c_tests_1.txt | cut -c1-300
timeout 300 python tools/cl_prof.py c4 > $out/r6c_cluster_phases_c4.txt 2>&1; cat $out/r6c_cluster_phases_c4.txt | cut -c1-400
tools/gpu_call.sh r6c "ab:c4:X=1:$V" "ab:c3:X=1:$V" "ab:c5:X=1:$V" "timeline:c4" "timeline:c5"
