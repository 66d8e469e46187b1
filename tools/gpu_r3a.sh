#!/bin/bash
# round 3, GPU call a: packed planes in rec_*_x6p -- parity first, then same-box A/B against the three-MFMA build
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
V=tools/probes/variants
timeout 900 python -m pytest tests/test_gpu_config_parity.py tests/test_reference_layers.py tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r3a_tests1.txt 2>&1
tail -4 gpurun_out/r3a_tests1.txt
tools/bench_variants.sh r3a "SBR_LIB=$V/libsbr_nopack.so" "SBR_DUMMY=1" "SBR_LIB=$V/libsbr_la2.so" "SBR_LIB=$V/libsbr_ns3.so" "SBR_TAIL_OVERLAP=0" "SBR_TAIL_OVERLAP=0 SBR_LIB=$V/libsbr_nopack.so" 2>&1 | tee gpurun_out/r3a_variants.txt
for lib in "" $V/libsbr_nopack.so; do
  echo "== rec_prof SBR_LIB=$lib"; SBR_LIB=$lib timeout 120 python tools/rec_prof.py c2 2>&1 | tail -25
  echo "== tail_prof SBR_LIB=$lib"; SBR_LIB=$lib timeout 120 python tools/tail_prof.py 2>&1 | tail -25
done > gpurun_out/r3a_prof.txt 2>&1
cat gpurun_out/r3a_prof.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r3a_tests_all.txt 2>&1
tail -4 gpurun_out/r3a_tests_all.txt
