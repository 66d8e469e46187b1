#!/bin/bash
# round 3, GPU call c: full GPU suite on the new default (packed planes + deferred forward stores), timeline, backward variants
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
V=tools/probes/variants
tools/bench_variants.sh r3c "SBR_DUMMY=1" "SBR_LIB=$V/libsbr_bdef.so" "SBR_LIB=$V/libsbr_bdefla2.so" "SBR_LIB=$V/libsbr_la2.so" "SBR_LIB=$V/libsbr_tok0.so" "SBR_LIB=$V/libsbr_tok2.so" "SBR_X6_PIPE=1" "SBR_TAIL_OVERLAP=0 SBR_LIB=$V/libsbr_bdefla2.so" "SBR_TAIL_SCATTER_WGS=128" "SBR_TAIL_SMALL_K=64" "SBR_TAIL_SMALL_K=32 SBR_TAIL_SMALL_SLABS=96" "SBR_DUMMY=2" 2>&1 | tee gpurun_out/r3c_variants.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/r3c_stats -o s -- python $OLDPWD/bench.py --steps 8 --warmup 3 --repeats 1 --no-cpu-baseline > $OLDPWD/gpurun_out/r3c_stats.log 2>&1 )
f=$(ls gpurun_out/r3c_stats/*/*kernel_trace.csv gpurun_out/r3c_stats/*kernel_trace.csv 2>/dev/null | head -1)
python tools/trace_gaps.py $f 3 > gpurun_out/r3c_timeline.txt 2>&1; cat gpurun_out/r3c_timeline.txt
SBR_LIB=$V/libsbr_bdefla2.so timeout 600 python -m pytest tests/test_gpu_config_parity.py tests/test_reference_layers.py tests/test_gpu_parity.py -m gpu -x -q -k "c2 or c1 or reference or overlapped or ring or pipelined" > gpurun_out/r3c_tests_bdef.txt 2>&1
tail -4 gpurun_out/r3c_tests_bdef.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r3c_tests_all.txt 2>&1
tail -25 gpurun_out/r3c_tests_all.txt
