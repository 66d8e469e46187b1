#!/bin/bash
# round 3, GPU call i: synchronous schedule of the 128-unit chains (one barrier per step) against the counter / gate schedule
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
V=tools/probes/variants
SBR_LIB=$V/libsbr_sync.so timeout 600 python -m pytest tests/test_gpu_config_parity.py tests/test_reference_layers.py tests/test_gpu_parity.py -m gpu -x -q -k "c2 or reference or one_layer or pipelined or overlapped or smallest or ragged" > gpurun_out/r3i_tests_sync.txt 2>&1; tail -5 gpurun_out/r3i_tests_sync.txt
tools/bench_variants.sh r3i "SBR_DUMMY=1" "SBR_LIB=$V/libsbr_sync.so" "SBR_DUMMY=2" "SBR_LIB=$V/libsbr_syncla2.so" "SBR_TAIL_OVERLAP=0" "SBR_TAIL_OVERLAP=0 SBR_LIB=$V/libsbr_sync.so" 2>&1 | tee gpurun_out/r3i_variants.txt
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r3i_tests_all.txt 2>&1; tail -5 gpurun_out/r3i_tests_all.txt
