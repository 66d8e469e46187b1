#!/bin/bash
# round 3, GPU call b: probes for the one-wave-per-SIMD question + forward variants
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
V=tools/probes/variants
timeout 120 tools/probes/own_valu_probe > gpurun_out/r3b_probe.txt 2>&1; cat gpurun_out/r3b_probe.txt
tools/bench_variants.sh r3b "SBR_DUMMY=1" "SBR_LIB=$V/libsbr_fv1.so" "SBR_LIB=$V/libsbr_fv2.so" "SBR_LIB=$V/libsbr_fv3.so" "SBR_LIB=$V/libsbr_fv7.so" "SBR_LIB=$V/libsbr_nop1.so" "SBR_LIB=$V/libsbr_nop3.so" "SBR_LIB=$V/libsbr_nop7.so" "SBR_LIB=$V/libsbr_la2.so" "SBR_DUMMY=2" 2>&1 | tee gpurun_out/r3b_variants.txt
SBR_LIB=$V/libsbr_fv7.so timeout 600 python -m pytest tests/test_gpu_config_parity.py tests/test_reference_layers.py -m gpu -x -q -k "c2 or c1 or l128 or reference or lstm128" > gpurun_out/r3b_tests_fv7.txt 2>&1
tail -4 gpurun_out/r3b_tests_fv7.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3b_tests_all.txt 2>&1
tail -6 gpurun_out/r3b_tests_all.txt
